"""Drop-in mirror of the reference's ``tacotron.model`` surface for the inference path.

Reference: tacotron/model.py -- ``Mode`` (:20-23), ``Tacotron(inputs, mode, training_summary)``
(:35-112) with attributes ``inp_sentences``, ``output_mel_spec`` (B,T,80),
``reduced_output_mel_spec`` (B,T/r,80*r), ``output_linear_spec`` (B,T,1025),
``alignment_history`` (T/r,B,T_sent) (:80-103,331,380-401), ``Tacotron.model_placeholders()``
(:606-656) and the PREDICT-mode dumps of ``summary()`` (:552-598).

There is no TensorFlow graph here: the "tensors" are symbolic :class:`Fetch` handles and
``Tacotron.run(fetches, feed_dict)`` plays the role of ``session.run`` (reference
tacotron/inference.py:75-85) by driving the HIP library.  Only ``Mode.PREDICT`` exists on this
path; TRAIN / EVAL are outside the accelerated scope and raise ``NotImplementedError``.
"""
import os

import numpy as np

from .._hip import Engine
from .params import inference_params, model_params


class Mode:
    TRAIN = 'train'
    EVAL = 'eval'
    PREDICT = 'predict'


class Placeholder(object):
    """Stand-in for tf.placeholder: a named feed slot."""

    def __init__(self, name, dtype, shape=None):
        self.name, self.dtype, self.shape = name, dtype, shape

    def __repr__(self):
        return 'Placeholder({!r})'.format(self.name)


class Fetch(object):
    """Stand-in for an output tf.Tensor: a named fetch."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return 'Fetch({!r})'.format(self.name)


class Tacotron(object):
    """Tacotron in PREDICT mode on one MI355X.

    ``weights`` ({tf variable name: array}, see tacotron/weights.py) may be given here or later
    through :meth:`restore`; it replaces ``tf.train.Saver().restore`` (reference
    tacotron/inference.py:55,71)."""

    def __init__(self, inputs, mode, training_summary=True, weights=None, hparams=None, device_id=0,
                 stream=None, engine=None):
        if mode != Mode.PREDICT:
            raise NotImplementedError('only Mode.PREDICT is implemented on the MI355X path '
                                      '(training / evaluation are out of scope)')
        self.hparams = hparams or model_params
        self._mode = mode
        self._training_summary = training_summary
        self.inp_sentences = inputs['ph_sentences']
        self.seq_lengths = inputs.get('ph_sentence_length')
        self.inp_mel_spec = inputs.get('ph_mel_specs')
        self.inp_linear_spec = inputs.get('ph_lin_specs')
        self.inp_time_steps = inputs.get('ph_time_frames')
        self.loss_op = self.loss_op_decoder = self.loss_op_post_processing = None
        self.output_mel_spec = Fetch('output_mel_spec')
        self.reduced_output_mel_spec = Fetch('reduced_output_mel_spec')
        self.output_linear_spec = Fetch('output_linear_spec')
        self.alignment_history = Fetch('alignment_history')
        self._summary = Fetch('summary')
        # `engine`: an Engine that already holds the weights (a second view of one handle, e.g. bench.py)
        self.engine = engine if engine is not None else Engine(self.hparams, device_id=device_id, stream=stream)
        self._loaded = engine is not None
        if weights is not None:
            self.restore(weights)

    def is_training(self):
        return self._mode == Mode.TRAIN

    # ------------------------------------------------------------------ weights
    def restore(self, weights):
        """weights: dict of arrays, a flat float32 blob in manifest order, a path to an ``.npz`` holding
        the manifest's variable names, or a TensorFlow checkpoint prefix / run directory."""
        if isinstance(weights, str):
            if weights.endswith('.npz'):
                with np.load(weights) as z:
                    weights = {k: z[k] for k in z.files}
            else:   # a TensorFlow checkpoint prefix or run directory (tacotron/inference.py:44-55,71)
                from .checkpoint import load_checkpoint
                weights = load_checkpoint(weights, self.hparams)
        if isinstance(weights, dict):
            self.engine.load_weights(weights)
        else:
            self.engine.load_weights_blob(weights)
        self._loaded = True

    # ------------------------------------------------------------------ execution
    def n_steps(self):
        # reference tacotron/model.py:309: maximum_iterations // reduction
        return self.hparams.decoder.maximum_iterations // self.hparams.reduction

    def predict_device(self, sentences, n_steps=None):
        """Runs encoder -> decoder -> post-net; returns device arrays."""
        sentences = np.ascontiguousarray(sentences, dtype=np.int32)
        if sentences.ndim != 2:
            raise ValueError('sentences must be (B, T_sent) int32')
        S = n_steps or self.n_steps()
        eng = self.engine
        memory = eng.encoder_forward(sentences)
        mel, align = eng.decoder_forward(memory, S)
        B = sentences.shape[0]
        mel.shape = (B, S * self.hparams.reduction, self.hparams.n_mels)  # model.py:383 reshape
        # (apply_post_processing=False, reference tacotron/model.py:388-391: the engine was created with the flag, its
        #  postnet_forward is then the final Dense alone, kernel (n_mels, 1 + n_fft / 2))
        linear = eng.postnet_forward(mel)
        return dict(memory=memory, mel=mel, alignments=align, linear=linear, n_steps=S)

    def run(self, fetches, feed_dict, n_steps=None):
        """session.run analogue: ``model.run([model.output_linear_spec], {model.inp_sentences: ids})``."""
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        sentences = None
        for k, v in feed_dict.items():
            if k is self.inp_sentences or getattr(k, 'name', None) == getattr(self.inp_sentences, 'name', object()):
                sentences = v
        if sentences is None:
            raise KeyError('feed_dict must feed model.inp_sentences')
        out = self.predict_device(sentences, n_steps)
        B = np.asarray(sentences).shape[0]
        S, r, nm = out['n_steps'], self.hparams.reduction, self.hparams.n_mels
        res = []
        for f in fl:
            if f.name == 'output_linear_spec':
                res.append(out['linear'].to_host())
            elif f.name == 'output_mel_spec':
                res.append(out['mel'].to_host())
            elif f.name == 'reduced_output_mel_spec':
                res.append(out['mel'].to_host().reshape(B, S, r * nm))
            elif f.name == 'alignment_history':
                res.append(out['alignments'].to_host())
            elif f.name == 'summary':
                res.append(self._dump(out))
            else:
                raise KeyError(f)
        return res[0] if single else res

    # ------------------------------------------------------------------ PREDICT-mode dumps
    def summary(self):
        """reference tacotron/model.py:453-604 -- in PREDICT mode the only effect of the summary op
        is the two .npz dumps (:555-598); fetch it through :meth:`run` to write them."""
        return self._summary

    def _dump(self, out):
        wrote = []
        if inference_params.dump_alignments:
            # (T/r, B, T_sent) -> (B, T_sent, T/r), key 'alignments'   (model.py:552,568)
            align = np.transpose(out['alignments'].to_host(), (1, 2, 0))
            path = os.path.join(inference_params.synthesis_dir, 'alignments.npz')
            np.savez(path, alignments=align)
            wrote.append(path)
        if inference_params.dump_linear_spectrogram:
            # batch item 0 as (1, 1025, T, 1), key 'linear_spec'      (model.py:584-596)
            spec = out['linear'].to_host()[0]
            path = os.path.join(inference_params.synthesis_dir, 'linear-spectrogram.npz')
            np.savez(path, linear_spec=spec.T[None, :, :, None])
            wrote.append(path)
        return wrote

    @staticmethod
    def model_placeholders():
        """reference tacotron/model.py:606-656: only ``ph_sentences`` is fed at inference."""
        return {
            'ph_sentences': Placeholder('ph_inp_sentences', np.int32, (None, None)),
            'ph_sentence_length': Placeholder('ph_sentence_length', np.int32),
            'ph_mel_specs': Placeholder('ph_mel_specs', np.float32),
            'ph_lin_specs': Placeholder('ph_lin_specs', np.float32),
            'ph_time_frames': Placeholder('ph_time_frames', np.int32),
        }
