"""ctypes binding of libsstts_hip.so (the C ABI declared in include/sstts_hip.h).

This is the only compute path of the package: there is no CPU fallback.  If the shared
library is missing or a call fails, an exception is raised.

Host arrays are numpy; device memory is either owned by the library's allocator
(:class:`DeviceArray`) or borrowed from anything exposing ``data_ptr()`` (a CUDA/HIP torch
tensor) -- PyTorch is optional plumbing, not a dependency of this module.
"""
import ctypes
import os
from ctypes import (POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64,
                    c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SSTTS_HIP_LIB') or os.path.join(_HERE, 'libsstts_hip.so')

TTS_OK = 0
TTS_ERR_INVALID = -1
TTS_ERR_NOT_LOADED = -2
TTS_ERR_HIP = -3
TTS_ERR_DB_RANGE = -4
TTS_ERR_UNSUPPORTED = -5


class TtsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('sstts_hip error {}: {}'.format(code, msg))
        self.code = code


class TtsConfig(ctypes.Structure):
    """struct tts_config (include/sstts_hip.h)."""
    _fields_ = [
        ('struct_size', c_int32),
        ('vocabulary_size', c_int32), ('embedding_size', c_int32), ('enc_prenet_units', c_int32 * 2),
        ('enc_n_banks', c_int32), ('enc_n_filters', c_int32), ('enc_proj_filters', c_int32 * 2),
        ('post_n_banks', c_int32), ('post_n_filters', c_int32), ('post_proj_filters', c_int32 * 2),
        ('n_highway_layers', c_int32), ('n_highway_units', c_int32), ('n_gru_units', c_int32),
        ('dec_prenet_units', c_int32 * 2), ('n_attention_units', c_int32),
        ('n_decoder_gru_units', c_int32), ('n_decoder_gru_layers', c_int32), ('n_mels', c_int32),
        ('reduction', c_int32), ('n_fft', c_int32), ('force_cudnn', c_int32),
        ('attention_mechanism', c_int32), ('luong_local_window_d', c_int32), ('luong_force_gaussian', c_int32),
        ('luong_local_mode', c_int32), ('apply_post_processing', c_int32),
    ]


class TtsSynthParams(ctypes.Structure):
    """struct tts_synth_params (include/sstts_hip.h)."""
    _fields_ = [
        ('n_steps', c_int32), ('ref_db', c_float), ('max_db', c_float), ('power', c_float),
        ('n_iter', c_int32), ('win_length', c_int32), ('hop_length', c_int32), ('seed', c_uint64),
        ('peak_normalize', c_int32), ('host_outputs', c_int32),
    ]


_PROTOTYPES = {
    'tts_version': (c_char_p, []),
    'tts_default_config': (c_int, [POINTER(TtsConfig)]),
    'tts_create': (c_int, [POINTER(TtsConfig), c_int, POINTER(c_void_p)]),
    'tts_destroy': (c_int, [c_void_p]),
    'tts_last_error': (c_char_p, [c_void_p]),
    'tts_set_stream': (c_int, [c_void_p, c_void_p]),
    'tts_set_option': (c_int, [c_void_p, c_char_p, c_int]),
    'tts_synchronize': (c_int, [c_void_p]),
    'tts_manifest_size': (c_int, [c_void_p]),
    'tts_manifest_entry': (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int64), POINTER(c_int)]),
    'tts_set_weight': (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    'tts_load_weights_blob': (c_int, [c_void_p, c_void_p, c_size_t]),
    'tts_finalize_weights': (c_int, [c_void_p]),
    'tts_malloc': (c_int, [POINTER(c_void_p), c_size_t]),
    'tts_free': (c_int, [c_void_p]),
    'tts_device_malloc': (c_int, [c_void_p, POINTER(c_void_p), c_size_t]),
    'tts_device_free': (c_int, [c_void_p, c_void_p]),
    'tts_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'tts_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'tts_memset': (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    'tts_encoder_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'tts_decoder_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tts_postnet_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'tts_denorm_power': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    'tts_griffin_lim': (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_void_p, c_void_p]),
    'tts_peak_normalize': (c_int, [c_void_p, c_void_p, c_int, c_int]),
    'tts_stft': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'tts_db_convert': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_void_p]),
    'tts_stft_magnitude': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'tts_mel_spectrogram': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                    c_void_p]),
    'tts_synthesize': (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(TtsSynthParams), c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    'tts_synthesize_host': (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(TtsSynthParams), POINTER(c_int)]),
    'tts_wait_host': (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_size_t)]),
    'tts_wait_host_outputs': (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_void_p), POINTER(c_size_t)]),
    'tts_profile_reset': (c_int, [c_void_p]),
    'tts_profile_get': (c_int, [c_void_p, c_char_p, POINTER(c_float), POINTER(c_int64)]),
    'tts_decoder_kernel_choice': (c_int, [c_void_p, c_int, c_int, c_int]),
    'tts_debug_workspace': (c_int, [c_void_p, c_char_p, POINTER(c_void_p), POINTER(c_size_t)]),
    'tts_debug_hold': (c_int, [c_void_p, c_int, c_int, ctypes.c_double]),
    'tts_debug_gemm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    'tts_debug_gl_plan': (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int), c_int, POINTER(c_int)]),
    'tts_device_info': (c_int, [c_void_p, c_char_p, POINTER(c_int)]),
}

_lib = None


def exported_symbols():
    """Names every include/sstts_hip.h entry point must resolve to."""
    return sorted(_PROTOTYPES)


def load_library(path=None):
    """dlopen the HIP library (no GPU needed for this) and attach prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise OSError('{} not found: build it with `python single-speaker-tts_amd/build.py` '
                      '(there is no CPU fallback)'.format(path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class DeviceArray(object):
    """A typed device buffer owned by the library allocator."""

    def __init__(self, engine, shape, dtype=np.float32):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = c_void_p()
        # allocated on the engine's device, whatever device is current on this thread
        rc = engine.lib.tts_device_malloc(engine.handle, byref(p), self.nbytes)
        if rc != TTS_OK:
            raise TtsError(rc, 'tts_device_malloc({}) failed'.format(self.nbytes))
        self.ptr = p.value

    def data_ptr(self):
        return self.ptr

    def copy_from(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        self.engine._check(self.engine.lib.tts_memcpy_h2d(self.engine.handle, self.ptr, host.ctypes.data,
                                                          self.nbytes))
        return self

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.engine._check(self.engine.lib.tts_memcpy_d2h(self.engine.handle, out.ctypes.data, self.ptr,
                                                          self.nbytes))
        return out

    def free(self):
        if self.ptr:
            if getattr(self.engine, 'handle', None):
                self.engine.lib.tts_device_free(self.engine.handle, self.ptr)
            else:   # the engine is gone (tts_destroy does not free caller-owned buffers)
                self.engine.lib.tts_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _is_device(x):
    return hasattr(x, 'data_ptr') and not isinstance(x, np.ndarray)


class Engine(object):
    """One handle = one GPU + one stream.  Mirrors the C ABI one to one."""

    def __init__(self, hparams=None, device_id=0, stream=None):
        self.lib = load_library()
        cfg = TtsConfig()
        self.lib.tts_default_config(byref(cfg))
        if hparams is not None:
            enc, dec, post = hparams.encoder, hparams.decoder, hparams.post
            cfg.vocabulary_size = hparams.vocabulary_size
            cfg.embedding_size = enc.embedding_size
            cfg.enc_prenet_units[0], cfg.enc_prenet_units[1] = [l[0] for l in enc.pre_net_layers]
            cfg.enc_n_banks, cfg.enc_n_filters = enc.n_banks, enc.n_filters
            cfg.enc_proj_filters[0], cfg.enc_proj_filters[1] = [p[0] for p in enc.projections]
            cfg.post_n_banks, cfg.post_n_filters = post.n_banks, post.n_filters
            cfg.post_proj_filters[0], cfg.post_proj_filters[1] = [p[0] for p in post.projections]
            cfg.n_highway_layers, cfg.n_highway_units = enc.n_highway_layers, enc.n_highway_units
            cfg.n_gru_units = enc.n_gru_units
            cfg.dec_prenet_units[0], cfg.dec_prenet_units[1] = [l[0] for l in dec.pre_net_layers]
            cfg.n_attention_units = dec.n_attention_units
            cfg.n_decoder_gru_units = dec.n_decoder_gru_units
            cfg.n_decoder_gru_layers = dec.n_gru_layers
            cfg.n_mels, cfg.reduction, cfg.n_fft = hparams.n_mels, hparams.reduction, hparams.n_fft
            cfg.force_cudnn = 1 if hparams.force_cudnn else 0
            att = hparams.attention
            if att.mechanism not in ('LuongAttention', 'LocalLuongAttention'):
                raise NotImplementedError('attention mechanism {!r}'.format(att.mechanism))
            if att.mechanism == 'LocalLuongAttention' and (att.luong_local_mode not in ('monotonic', 'predictive') or
                                                           att.luong_local_score != 'dot'):
                raise NotImplementedError('LocalLuongAttention: the general / concat scores raise '
                                          'NotImplementedError in the reference too')
            cfg.luong_local_mode = 1 if att.luong_local_mode == 'predictive' else 0
            cfg.attention_mechanism = 1 if att.mechanism == 'LocalLuongAttention' else 0
            cfg.luong_local_window_d = att.luong_local_window_D
            cfg.luong_force_gaussian = 1 if att.luong_force_gaussian else 0
            cfg.apply_post_processing = 1 if hparams.apply_post_processing else 0
        self.cfg = cfg
        h = c_void_p()
        rc = self.lib.tts_create(byref(cfg), device_id, byref(h))
        if rc != TTS_OK:
            raise TtsError(rc, (self.lib.tts_last_error(None) or b'').decode())
        self.handle = h
        self.device_id = int(device_id)
        self._staging = {}
        if stream is not None:
            self._check(self.lib.tts_set_stream(self.handle, c_void_p(stream)))

    def set_stream(self, stream):
        """Adopt a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); None: a stream of the library's own."""
        self._check(self.lib.tts_set_stream(self.handle, c_void_p(stream) if stream else None))

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != TTS_OK:
            msg = (self.lib.tts_last_error(self.handle) or b'').decode()
            if rc == TTS_ERR_DB_RANGE:
                raise AssertionError(msg)   # reference audio/conversion.py:47-49
            raise TtsError(rc, msg)

    def close(self):
        if getattr(self, 'handle', None):
            for d in self._staging.values():
                d.free()
            self._staging = {}
            self.lib.tts_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._check(self.lib.tts_set_option(self.handle, key.encode(), int(value)))

    def synchronize(self):
        self._check(self.lib.tts_synchronize(self.handle))

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def to_device(self, host, dtype=None):
        host = np.asarray(host)
        return DeviceArray(self, host.shape, dtype or host.dtype).copy_from(host)

    def _in(self, x, dtype, role=None):
        """-> (pointer, keepalive) for a host array or a device buffer.

        Host arrays are uploaded with tts_memcpy_h2d, which waits for the handle's stream: a call fed from
        host memory therefore starts after the previous call's Griffin-Lim has finished, i.e. the stream
        pipelining of tts_synthesize only overlaps calls whose inputs are device resident.  With a `role`
        the staging buffer is kept and reused by later calls of the same size (a fresh buffer per call
        would add a hipFree, which synchronises the whole device)."""
        if x is None:
            return None, None
        if _is_device(x):
            return x.data_ptr(), x
        host = np.ascontiguousarray(x, dtype=dtype)
        if role is None:
            d = self.to_device(host, dtype)
            return d.ptr, d
        d = self._staging.get(role)
        if d is None or d.nbytes != host.nbytes:
            if d is not None:
                d.free()
            d = DeviceArray(self, host.shape, dtype)
            self._staging[role] = d
        d.shape = host.shape
        d.copy_from(host)
        return d.ptr, d

    def _check_ids(self, ids):
        """tf.nn.embedding_lookup on the CPU raises for ids outside the table (reference
        tacotron/model.py:154); device-resident ids cannot be inspected without a synchronisation, the
        kernel reads them as a zero row (TF's GPU behaviour)."""
        if isinstance(ids, np.ndarray) and ids.size:
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= self.cfg.vocabulary_size:
                raise TtsError(TTS_ERR_INVALID, 'sentence ids must lie in [0, {}): got [{}, {}]'.format(
                    self.cfg.vocabulary_size, lo, hi))

    # ------------------------------------------------------------------ weights
    def manifest(self):
        out = []
        n = self.lib.tts_manifest_size(self.handle)
        for i in range(n):
            name = c_char_p()
            shape = (c_int64 * 4)()
            nd = c_int()
            self._check(self.lib.tts_manifest_entry(self.handle, i, byref(name), shape, byref(nd)))
            out.append((name.value.decode(), tuple(int(shape[d]) for d in range(nd.value))))
        return out

    def load_weights(self, weights):
        """weights: {tf variable name: array in TensorFlow layout}."""
        for name, shape in self.manifest():
            if name not in weights:
                raise TtsError(TTS_ERR_NOT_LOADED, 'missing weight ' + name)
            w = np.ascontiguousarray(weights[name], dtype=np.float32)
            shp = (c_int64 * max(1, w.ndim))(*w.shape)
            self._check(self.lib.tts_set_weight(self.handle, name.encode(), w.ctypes.data, shp, w.ndim))
        self._check(self.lib.tts_finalize_weights(self.handle))

    def load_weights_blob(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32).reshape(-1)
        self._check(self.lib.tts_load_weights_blob(self.handle, blob.ctypes.data, blob.size))
        self._check(self.lib.tts_finalize_weights(self.handle))

    # ------------------------------------------------------------------ stages
    def encoder_forward(self, ids, out=None):
        B, Ts = ids.shape
        self._check_ids(ids)
        p_ids, _k = self._in(ids, np.int32, 'ids')
        mem = out if out is not None else self.empty((B, Ts, 2 * self.cfg.n_gru_units))
        self._check(self.lib.tts_encoder_forward(self.handle, p_ids, B, Ts, mem.data_ptr()))
        return mem

    def decoder_forward(self, memory, n_steps, want_alignments=True, mel=None, alignments=None):
        B, Ts = memory.shape[0], memory.shape[1]
        p_mem, _k = self._in(memory, np.float32)
        if mel is None:
            mel = self.empty((B, n_steps, self.cfg.reduction * self.cfg.n_mels))
        if alignments is None and want_alignments:
            alignments = self.empty((n_steps, B, Ts))
        self._check(self.lib.tts_decoder_forward(self.handle, p_mem, B, Ts, n_steps, mel.data_ptr(),
                                                 alignments.data_ptr() if alignments is not None else None))
        return mel, alignments

    def postnet_forward(self, mel, out=None):
        B, T = mel.shape[0], mel.shape[1]
        p_mel, _k = self._in(mel, np.float32)
        lin = out if out is not None else self.empty((B, T, 1 + self.cfg.n_fft // 2))
        self._check(self.lib.tts_postnet_forward(self.handle, p_mel, B, T, lin.data_ptr()))
        return lin

    def denorm_power(self, linear, ref_db, max_db, power, out=None):
        B, T, F = linear.shape
        p_lin, _k = self._in(linear, np.float32)
        mag = out if out is not None else self.empty((B, F, T))
        self._check(self.lib.tts_denorm_power(self.handle, p_lin, B, T, F, ref_db, max_db, power, mag.data_ptr()))
        return mag

    def griffin_lim(self, mag, n_iter, win_length, hop_length, n_fft, init_phase=None, seed=0, want_mse=True):
        B, F, T = mag.shape
        p_mag, _k1 = self._in(mag, np.float32)
        p_init, _k2 = self._in(init_phase, np.float32)
        wav = self.empty((B, hop_length * (T - 1)))
        mse = self.empty((B,)) if want_mse else None
        self._check(self.lib.tts_griffin_lim(self.handle, p_mag, p_init, seed, B, T, n_iter, win_length, hop_length,
                                             n_fft, wav.data_ptr(), mse.data_ptr() if mse is not None else None))
        return wav, mse

    def peak_normalize(self, wav):
        B, n = wav.shape
        self._check(self.lib.tts_peak_normalize(self.handle, wav.data_ptr(), B, n))
        return wav

    def synthesize(self, ids, n_steps, ref_db, max_db, power, n_iter, win_length, hop_length, init_phase=None,
                   seed=0, peak_normalize=True, want_mel=False, want_alignments=False, want_linear=False, wav=None):
        B, Ts = ids.shape
        T = n_steps * self.cfg.reduction
        F = 1 + self.cfg.n_fft // 2
        sp = TtsSynthParams(n_steps, ref_db, max_db, power, n_iter, win_length, hop_length, seed,
                            1 if peak_normalize else 0)
        self._check_ids(ids)
        p_ids, _k1 = self._in(ids, np.int32, 'ids')
        p_init, _k2 = self._in(init_phase, np.float32, 'init_phase')
        wav = wav if wav is not None else self.empty((B, hop_length * (T - 1)))
        # want_*: False, True (a fresh buffer) or a device array of the right size to write into (no allocation in the call)
        def _out(want, shape):
            if want is None or want is False:
                return None
            if want is True:
                return self.empty(shape)
            if int(np.prod(want.shape)) != int(np.prod(shape)):
                raise ValueError('synthesize: output buffer of shape {} given, {} needed'.format(want.shape, shape))
            return want
        mel = _out(want_mel, (B, T, self.cfg.n_mels))
        ali = _out(want_alignments, (n_steps, B, Ts))
        lin = _out(want_linear, (B, T, F))
        self._check(self.lib.tts_synthesize(self.handle, p_ids, B, Ts, byref(sp), p_init, wav.data_ptr(),
                                            mel.data_ptr() if mel is not None else None,
                                            ali.data_ptr() if ali is not None else None,
                                            lin.data_ptr() if lin is not None else None))
        return dict(wav=wav, mel=mel, alignments=ali, linear=lin)

    def synthesize_host(self, ids, n_steps, ref_db, max_db, power, n_iter, win_length, hop_length, seed=0,
                        peak_normalize=True, want_linear=False, want_alignments=False):
        """Asynchronous end-to-end call on HOST ids (int32 (B, T_sent)): returns a ticket at once; the upload, the
        network, Griffin-Lim and the download of the waveforms into pinned memory overlap with the neighbouring
        calls.  Keep at most three calls in flight: submit k + 2, then ``wait_host(ticket_k)``."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self._check_ids(ids)
        B, Ts = ids.shape
        sp = TtsSynthParams(n_steps, ref_db, max_db, power, n_iter, win_length, hop_length, seed, 1 if peak_normalize else 0,
                            (1 if want_linear else 0) | (2 if want_alignments else 0))
        t = c_int(-1)
        self._check(self.lib.tts_synthesize_host(self.handle, ids.ctypes.data, B, Ts, byref(sp), byref(t)))
        self._host_shapes = getattr(self, '_host_shapes', {})
        self._host_shapes[t.value] = (B, hop_length * (n_steps * self.cfg.reduction - 1))
        self._host_out_shapes = getattr(self, '_host_out_shapes', {})
        self._host_out_shapes[t.value] = ((B, n_steps * self.cfg.reduction, 1 + self.cfg.n_fft // 2), (n_steps, B, Ts))
        return t.value

    def wait_host_outputs(self, ticket, copy=True):
        """(linear (B, T, F) or None, alignments (n_steps, B, Ts) or None) of a ``synthesize_host`` call made with
        ``want_linear`` / ``want_alignments``; views of pinned buffers with the waveform buffer's lifetime unless ``copy``.
        Call it BEFORE ``wait_host`` of the same ticket or keep the shapes yourself (it does not consume the ticket)."""
        pl, pa = c_void_p(), c_void_p()
        nl, na = c_size_t(0), c_size_t(0)
        self._check(self.lib.tts_wait_host_outputs(self.handle, int(ticket), byref(pl), byref(nl), byref(pa), byref(na)))
        shl, sha = self._host_out_shapes.pop(ticket, ((nl.value,), (na.value,)))
        lin = np.ctypeslib.as_array(ctypes.cast(pl, POINTER(c_float)), shape=(nl.value,)).reshape(shl) if nl.value else None
        ali = np.ctypeslib.as_array(ctypes.cast(pa, POINTER(c_float)), shape=(na.value,)).reshape(sha) if na.value else None
        if copy:
            lin = None if lin is None else lin.copy()
            ali = None if ali is None else ali.copy()
        return lin, ali

    def wait_host(self, ticket, copy=True):
        """Waveforms (B, hop*(T-1)) float32 of a ``synthesize_host`` call.  ``copy=False`` returns a view of the library's
        pinned buffer, valid until the THIRD ``synthesize_host`` call (three buffer sets) after the one that produced it."""
        p = c_void_p()
        n = c_size_t(0)
        self._check(self.lib.tts_wait_host(self.handle, int(ticket), byref(p), byref(n)))
        shape = self._host_shapes.pop(ticket, (n.value,))
        view = np.ctypeslib.as_array(ctypes.cast(p, POINTER(c_float)), shape=(n.value,)).reshape(shape)
        return view.copy() if copy else view

    def stft(self, wav, n_fft, win_length, hop_length):
        """complex64 (B, F, n_frames) = librosa.stft per utterance."""
        B, n = wav.shape
        p_wav, _k = self._in(wav, np.float32)
        Tf = 1 + n // hop_length
        out = self.empty((B, 1 + n_fft // 2, Tf), np.complex64)
        self._check(self.lib.tts_stft(self.handle, p_wav, B, n, n_fft, win_length, hop_length, out.data_ptr()))
        return out

    def stft_magnitude(self, wav, n_fft, win_length, hop_length, power=1.0):
        B, n = wav.shape
        p_wav, _k = self._in(wav, np.float32)
        Tf = 1 + n // hop_length
        out = self.empty((B, 1 + n_fft // 2, Tf))
        self._check(self.lib.tts_stft_magnitude(self.handle, p_wav, B, n, n_fft, win_length, hop_length, power,
                                                out.data_ptr()))
        return out

    def mel_spectrogram(self, lin, n_fft, sampling_rate, n_mels, fmin, fmax):
        B, F, Tf = lin.shape
        p_lin, _k = self._in(lin, np.float32)
        out = self.empty((B, n_mels, Tf))
        self._check(self.lib.tts_mel_spectrogram(self.handle, p_lin, B, Tf, n_fft, sampling_rate, n_mels, fmin,
                                                 fmax if fmax is not None else 0.0, out.data_ptr()))
        return out

    def db_convert(self, x, mode, ref_db=0.0, max_db=0.0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        d = self.to_device(x)
        self._check(self.lib.tts_db_convert(self.handle, d.ptr, x.size, mode, ref_db, max_db, d.ptr))
        return d.to_host()

    # ------------------------------------------------------------------ profiling / debug
    def profile_reset(self):
        self._check(self.lib.tts_profile_reset(self.handle))

    def decoder_kernel_choice(self, B, Ts, pipelined=True):
        """0 launch per layer, 1 persistent (streamed weights), 2 persistent (weight-stationary): what a call of this shape takes"""
        rc = self.lib.tts_decoder_kernel_choice(self.handle, int(B), int(Ts), 1 if pipelined else 0)
        if rc < 0:
            self._check(rc)
        return rc

    def device_info(self):
        """(uuid as 32 hex digits, compute units) of the handle's device"""
        buf = ctypes.create_string_buffer(33)
        n = c_int()
        self._check(self.lib.tts_device_info(self.handle, buf, byref(n)))
        return buf.value.decode(), n.value

    def profile_get(self, stage):
        ms = c_float()
        n = c_int64()
        self._check(self.lib.tts_profile_get(self.handle, stage.encode(), byref(ms), byref(n)))
        return ms.value, n.value

    def debug_workspace(self, name, shape, dtype=np.float32):
        p = c_void_p()
        nb = c_size_t()
        self._check(self.lib.tts_debug_workspace(self.handle, name.encode(), byref(p), byref(nb)))
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= nb.value, (name, out.nbytes, nb.value)
        self._check(self.lib.tts_memcpy_d2h(self.handle, out.ctypes.data, p, out.nbytes))
        return out
