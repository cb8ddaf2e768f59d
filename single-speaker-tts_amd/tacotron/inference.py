"""Drop-in mirror of the reference's ``tacotron.inference`` (tacotron/inference.py).

``pad_sentence`` (:22-27), ``inference(model, sentences)`` (:30-105) and the body of
``__main__`` (:130-200) as :func:`synthesize_sentences`: ids -> padded batch -> network ->
de-normalise -> ``** magnitude_power`` -> Griffin-Lim -> ``{i+1}.wav``.

Where the reference fans Griffin-Lim out to 6 worker processes, one utterance each
(:185-188), the whole batch is reconstructed by one batched kernel sequence on the GPU.
"""
import os

import numpy as np

from ..audio.conversion import ms_to_samples
from ..audio.io import save_wav
from .model import Mode, Tacotron
from .params import dataset_params, inference_params, model_params


def pad_sentence(_sentence, _max_len):
    """reference tacotron/inference.py:22-27."""
    pad_len = _max_len - len(_sentence)
    pad_token = dataset_params.vocabulary_dict['pad']
    return np.append(_sentence, [pad_token] * pad_len)


def inference(model, sentences, n_steps=None):
    """reference tacotron/inference.py:30-105.

    Arguments:
        model (Tacotron): model with restored weights.
        sentences: list/array of padded id sequences (B, T_sent).

    Returns:
        list of np.ndarray: per utterance the linear-scale magnitude spectrogram, shape
        (1025, T) float32 = decibel_to_magnitude(inv_normalize_decibel(spec.T, mel_ref_db,
        mel_max_db)) exactly as the reference (it uses the *mel* dB constants, :96-98).
    """
    sentences = np.asarray(sentences, dtype=np.int32)
    if inference_params.dump_alignments or inference_params.dump_linear_spectrogram:
        fetches = [model.summary(), model.output_linear_spec] \
            if os.path.isdir(inference_params.synthesis_dir) else [model.output_linear_spec]
    else:
        fetches = [model.output_linear_spec]
    out = model.predict_device(sentences, n_steps)
    if len(fetches) == 2:
        model._dump(out)
    loader = dataset_params.dataset_loader
    mag = model.engine.denorm_power(out['linear'], loader.mel_mag_ref_db, loader.mel_mag_max_db, 1.0).to_host()
    return [mag[b] for b in range(mag.shape[0])]


def synthesize_batch(model, sentences, n_steps=None, n_iter=None, init_phase=None, seed=0, peak_normalize=False):
    """ids (B, T_sent) -> waveforms (B, hop*(T-1)) float32: inference() + the synthesize() closure
    of the reference (tacotron/inference.py:170-188) fused into one device call."""
    hp = model.hparams
    loader = dataset_params.dataset_loader
    win_len = ms_to_samples(hp.win_len, hp.sampling_rate)
    win_hop = ms_to_samples(hp.win_hop, hp.sampling_rate)
    S = n_steps or model.n_steps()
    out = model.engine.synthesize(np.ascontiguousarray(sentences, dtype=np.int32), S, loader.mel_mag_ref_db,
                                  loader.mel_mag_max_db, hp.magnitude_power,
                                  hp.reconstruction_iterations if n_iter is None else n_iter, win_len, win_hop,
                                  init_phase=init_phase, seed=seed, peak_normalize=peak_normalize)
    return out['wav'].to_host()


def synthesize_stream(model, batches, n_steps=None, n_iter=None, seed=0, peak_normalize=False, copy=False, want_linear=False,
                      want_alignments=False):
    """Generator over batches of padded id sequences (each (B, T_sent) int32, HOST arrays) -> per batch the waveforms
    (B, hop*(T-1)) float32 in host memory, with THREE batches in flight: batch k + 2 is uploaded and encoded, batch k + 1
    is in its decoder, batch k in its post-net / Griffin-Lim while batch k - 1 is being downloaded (the reference runs the
    batches one after the other, tacotron/inference.py:75-101,185-200).  The arrays yielded are views of the library's
    pinned buffers unless ``copy``: valid until two more batches have been requested.

    With ``want_linear`` / ``want_alignments`` every item is a tuple ``(wavs, linear, alignments)``: the normalised linear
    spectrograms (B, T, 1025) -- what ``model.output_linear_spec`` is, the thing the reference's ``inference()`` fetches
    (:75-92) -- and the alignments (n_steps, B, T_sent) of the same call, downloaded behind the waveforms (None where not
    asked for)."""
    hp = model.hparams
    loader = dataset_params.dataset_loader
    win_len = ms_to_samples(hp.win_len, hp.sampling_rate)
    win_hop = ms_to_samples(hp.win_hop, hp.sampling_rate)
    S = n_steps or model.n_steps()
    it = hp.reconstruction_iterations if n_iter is None else n_iter
    eng = model.engine
    extra = want_linear or want_alignments

    def collect(ticket):
        if not extra:
            return eng.wait_host(ticket, copy=copy)
        lin, ali = eng.wait_host_outputs(ticket, copy=copy)
        return eng.wait_host(ticket, copy=copy), lin, ali

    # three batches in flight (the library's three buffer sets): the encoder of batch k + 2 runs one inter-Griffin-Lim gap
    # ahead of its decoder, which follows the decoder of batch k + 1 without a pause, beside the Griffin-Lim of batch k
    pending = []
    for k, ids in enumerate(batches):
        pending.append(eng.synthesize_host(ids, S, loader.mel_mag_ref_db, loader.mel_mag_max_db, hp.magnitude_power, it, win_len,
                                           win_hop, seed=seed + k, peak_normalize=peak_normalize, want_linear=want_linear,
                                           want_alignments=want_alignments))
        if len(pending) == 3:
            yield collect(pending.pop(0))
    while pending:
        yield collect(pending.pop(0))


def inference_stream(model, batches, n_steps=None, n_iter=None, seed=0):
    """``inference()`` over a stream of batches with three calls in flight: per batch ``(spectrograms, waveforms)`` where
    ``spectrograms`` is what the reference's ``inference()`` returns for that batch -- per utterance the (1025, T) linear
    magnitude spectrogram ``decibel_to_magnitude(inv_normalize_decibel(spec.T, mel_ref_db, mel_max_db))``
    (tacotron/inference.py:93-101; computed on the host from the downloaded network output with the conversions of
    ``audio.conversion``) -- and ``waveforms`` the Griffin-Lim reconstructions the reference's ``__main__`` makes of them
    (:170-188)."""
    loader = dataset_params.dataset_loader
    ref_db, max_db = np.float32(loader.mel_mag_ref_db), np.float32(loader.mel_mag_max_db)
    rng_db = np.float32(abs(float(ref_db)) + abs(float(max_db)))
    for wavs, lin, _ in synthesize_stream(model, batches, n_steps=n_steps, n_iter=n_iter, seed=seed, copy=True, want_linear=True):
        specs = []
        for b in range(lin.shape[0]):
            # inv_normalize_decibel, decibel_to_magnitude (reference audio/conversion.py:81-102, 32-53) in host arithmetic:
            # the device versions of audio.conversion would synchronise the stream that the next batch is running on
            db = (np.clip(lin[b].T, np.float32(0), np.float32(1)) - np.float32(1)) * rng_db + ref_db
            assert not (db < -100.0).any(), 'decibel_to_magnitude: values below -100 dB'
            specs.append(np.power(np.float32(10), db / np.float32(20)).astype(np.float32))
        yield specs, wavs


def synthesize_sentences(raw_sentences, weights, dataset=None, out_dir=None, device_id=0, seed=0):
    """The reference's ``__main__`` (tacotron/inference.py:130-200) as a function.

    raw text lines -> process_sentences -> pad -> model -> wavs -> ``{i+1}.wav`` (peak-normalised
    float32 WAV, save_wav(norm=True)).  Returns the list of waveforms."""
    from ..datasets.lj_speech import LJSpeechDatasetHelper
    out_dir = out_dir or inference_params.synthesis_dir
    if not os.path.isdir(out_dir):
        raise NotADirectoryError('The specified synthesis target folder does not exist.')
    dataset = dataset or LJSpeechDatasetHelper(dataset_folder=dataset_params.dataset_folder,
                                                char_dict=dataset_params.vocabulary_dict, fill_dict=False)
    id_sequences, sequence_lengths = dataset.process_sentences(raw_sentences)
    sentences = [np.frombuffer(s, dtype=np.int32) for s in id_sequences]
    max_length = max(sequence_lengths)
    sentences = np.array([pad_sentence(s, max_length) for s in sentences], dtype=np.int32)
    model = Tacotron(inputs=Tacotron.model_placeholders(), mode=Mode.PREDICT, weights=weights, device_id=device_id)
    wavs = synthesize_batch(model, sentences, seed=seed, peak_normalize=False)
    for i, wav in enumerate(wavs):
        save_wav(os.path.join(out_dir, '{}.wav'.format(i + 1)), wav, model_params.sampling_rate, True)
    return list(wavs)


def read_sentences(path):
    """reference tacotron/inference.py:139-143: one sentence per line, the newline removed (nothing else stripped)."""
    raw_sentences = []
    with open(path, 'r') as f_sent:
        for line in f_sent:
            raw_sentences.append(line.replace('\n', ''))
    return raw_sentences


def main(argv=None):
    """The reference's ``python tacotron/inference.py`` (tacotron/inference.py:130-200): read
    ``inference_params.synthesis_file``, restore the checkpoint named by ``inference_params`` (``checkpoint_file``, or the
    latest one of ``checkpoint_dir/checkpoint_load_run``: :44-55), synthesize, write ``{i+1}.wav`` into ``synthesis_dir``.

        python -m single-speaker-tts_amd.tacotron.inference [--synthesis-file F] [--synthesis-dir D]
                                                            [--weights CKPT | --synthetic-weights SEED]

    The options override the ``inference_params`` fields of the same name.  ``--weights`` takes what
    ``Tacotron.restore`` takes (a TensorFlow checkpoint prefix or run directory, or an ``.npz`` of the manifest's
    variables); ``--synthetic-weights`` a seed for the synthetic initialiser (no checkpoint ships with the reference)."""
    import argparse
    ap = argparse.ArgumentParser(prog='tacotron.inference')
    ap.add_argument('--synthesis-file', default=None)
    ap.add_argument('--synthesis-dir', default=None)
    ap.add_argument('--weights', default=None)
    ap.add_argument('--synthetic-weights', type=int, default=None)
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--seed', type=int, default=0, help='seed of the Griffin-Lim start phases (the reference draws them unseeded)')
    args = ap.parse_args(argv)
    out_dir = args.synthesis_dir or inference_params.synthesis_dir
    # Before we start doing anything we check if the required target folder actually exists (:131-133)
    if not os.path.isdir(out_dir):
        raise NotADirectoryError('The specified synthesis target folder does not exist.')
    raw_sentences = read_sentences(args.synthesis_file or inference_params.synthesis_file)
    print('{} sentences were loaded for inference.'.format(len(raw_sentences)))
    if args.synthetic_weights is not None:
        from .weights import synthetic_weights
        weights = synthetic_weights(args.synthetic_weights, model_params)
    elif args.weights is not None:
        weights = args.weights
    elif inference_params.checkpoint_file is not None:
        weights = inference_params.checkpoint_file
    else:
        weights = os.path.join(inference_params.checkpoint_dir, inference_params.checkpoint_load_run)
    wavs = synthesize_sentences(raw_sentences, weights, out_dir=out_dir, device_id=args.device, seed=args.seed)
    for i in range(len(wavs)):
        print('Saved: "{}"'.format(os.path.join(out_dir, '{}.wav'.format(i + 1))))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
