"""Mirror of the reference's serving helpers (tacotron/serve.py:22-126) on the MI355X path.

``pre_process_sentences`` (:22-36) and ``post_process_spectrograms`` (:39-86) keep their names;
``serve`` (:89-126) is the same loop -- sentences from a generator, one batch per iteration --
driving the HIP library instead of a TensorFlow SavedModel session.  Unlike the reference's
``post_process_spectrograms`` (which squeezes a trailing axis and therefore only works for a batch
of one, serve.py:58), a whole batch is supported.
"""
import numpy as np

from ..audio.conversion import ms_to_samples
from .inference import pad_sentence
from .model import Mode, Tacotron
from .params import dataset_params, model_params


def pre_process_sentences(_sentences, dataset):
    """raw strings -> padded int32 id batch (reference tacotron/serve.py:22-36)."""
    id_sequences, sequence_lengths = dataset.process_sentences(_sentences)
    sentences = [np.frombuffer(s, dtype=np.int32) for s in id_sequences]
    max_length = max(sequence_lengths)
    return np.array([pad_sentence(s, max_length) for s in sentences], dtype=np.int32)


def post_process_spectrograms(_spectrograms, engine, init_phase=None, seed=0):
    """normalised linear spectrograms (B, T, 1025) -> list of waveforms: de-normalise with the mel dB
    constants, ``** magnitude_power``, Griffin-Lim (reference tacotron/serve.py:39-86)."""
    loader = dataset_params.dataset_loader
    win_len = ms_to_samples(model_params.win_len, model_params.sampling_rate)
    win_hop = ms_to_samples(model_params.win_hop, model_params.sampling_rate)
    spec = np.asarray(_spectrograms, dtype=np.float32)
    if spec.ndim == 2:
        spec = spec[None]
    mag = engine.denorm_power(spec, loader.mel_mag_ref_db, loader.mel_mag_max_db, model_params.magnitude_power)
    wav, _ = engine.griffin_lim(mag, model_params.reconstruction_iterations, win_len, win_hop, model_params.n_fft,
                                init_phase=init_phase, seed=seed, want_mse=False)
    wav = wav.to_host()
    return [wav[b] for b in range(wav.shape[0])]


def serve(sentence_generator, weights, dataset=None, device_id=0, pipelined=False):
    """Generator: for each batch of raw sentences yield the list of synthesized waveforms
    (reference tacotron/serve.py:89-126, with the SavedModel session replaced by the engine).

    Default: the reference's request/response order -- every batch is answered before the next one is pulled from
    ``sentence_generator`` (reference serve.py:108-124 blocks on a live generator, and a client may wait for its answer
    before it sends more).  ``pipelined=True`` keeps three batches in flight for OFFLINE streams whose batches are all
    available: batch k is then yielded only after batch k + 2 has been pulled from the generator (the last one when the
    generator ends), which on a request-driven generator would hold every answer back by two requests."""
    from ..datasets.lj_speech import LJSpeechDatasetHelper
    dataset = dataset or LJSpeechDatasetHelper(dataset_folder=dataset_params.dataset_folder,
                                                char_dict=dataset_params.vocabulary_dict, fill_dict=False)
    model = Tacotron(inputs=Tacotron.model_placeholders(), mode=Mode.PREDICT, weights=weights, device_id=device_id)
    if not pipelined:   # the reference's loop as it stands: one batch at a time, spectrograms through host memory
        for sentences in sentence_generator:
            ids = pre_process_sentences(sentences, dataset)
            spectrograms = model.run(model.output_linear_spec, {model.inp_sentences: ids})
            yield post_process_spectrograms(spectrograms, model.engine)
        return
    # three batches in flight, nothing but ids and waveforms crosses the host boundary (inference.synthesize_stream)
    from .inference import synthesize_stream
    batches = (pre_process_sentences(sentences, dataset) for sentences in sentence_generator)
    for wavs in synthesize_stream(model, batches, peak_normalize=False, copy=True):
        yield [wavs[b] for b in range(wavs.shape[0])]
