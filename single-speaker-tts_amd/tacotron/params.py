"""Hyper-parameter singletons of the Tacotron inference path.

Mirrors the field names and default values of the reference's HParams
singletons (plain dataclasses; no TensorFlow):

  * ``model_params``      <- reference tacotron/params/model.py:8-153
  * ``inference_params``  <- reference tacotron/params/inference.py:4-35
  * ``dataset_params``    <- reference tacotron/params/dataset.py:9-32
  * LJ-Speech dB constants <- reference datasets/lj_speech.py:20-29

Activations are stored as strings ('relu' / None) instead of TF function
objects.  ``force_cudnn`` selects the GRU formulation (SURVEY.md S5 vs S5'):
the reference default is True (CudnnCompatibleGRUCell, GPU only in TF); the
parity target named by BASELINE.json is the TF *CPU* path, i.e. the plain
``GRUCell`` formulation, so the default here is False.  Both are implemented.
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple


@dataclass
class EncoderParams:
    embedding_size: int = 256
    # (units, dropout, activation); dropout is inactive outside Mode.TRAIN
    # (reference tacotron/model.py:119-122, layers.py:299-302).
    pre_net_layers: Tuple = ((256, 0.5, 'relu'), (128, 0.5, 'relu'))
    n_banks: int = 16
    n_filters: int = 128
    # (filters, kernel_size, activation)
    projections: Tuple = ((128, 3, 'relu'), (128, 3, None))
    n_highway_layers: int = 4
    n_highway_units: int = 128
    n_gru_units: int = 128


@dataclass
class DecoderParams:
    pre_net_layers: Tuple = ((256, 0.5, 'relu'), (128, 0.5, 'relu'))
    n_gru_layers: int = 2
    n_decoder_gru_units: int = 256
    n_attention_units: int = 256
    target_size: int = 80
    maximum_iterations: int = 1000


@dataclass
class AttentionParams:
    """model_params.attention (reference tacotron/params/model.py:112-128).  ``mechanism`` is the class
    name as a string; the enum values of AttentionMode / AttentionScore (reference
    tacotron/attention.py:14-29) are spelled as lower-case strings."""
    mechanism: str = 'LuongAttention'          # | 'LocalLuongAttention'
    luong_local_score: str = 'dot'             # 'general' / 'concat' raise NotImplementedError in the reference
    luong_local_mode: str = 'monotonic'        # or 'predictive' (p = S sigmoid(v_p^T tanh(W_p h)), attention.py:246-258)
    luong_force_gaussian: bool = True
    luong_local_window_D: int = 10


@dataclass
class PostParams:
    n_banks: int = 8
    n_filters: int = 128
    projections: Tuple = ((256, 3, 'relu'), (80, 3, None))
    n_highway_layers: int = 4
    n_highway_units: int = 128
    n_gru_units: int = 128


@dataclass
class ModelParams:
    vocabulary_size: int = 39
    sampling_rate: int = 22050
    n_fft: int = 2048
    win_len: float = 50.0     # ms
    win_hop: float = 12.5     # ms
    n_mels: int = 80
    mel_fmin: int = 0
    mel_fmax: int = 8000
    n_mfcc: int = 13
    reduction: int = 5
    apply_post_processing: bool = True
    magnitude_power: float = 1.3
    reconstruction_iterations: int = 50
    force_cudnn: bool = False
    encoder: EncoderParams = field(default_factory=EncoderParams)
    decoder: DecoderParams = field(default_factory=DecoderParams)
    attention: AttentionParams = field(default_factory=AttentionParams)
    post: PostParams = field(default_factory=PostParams)


@dataclass
class InferenceParams:
    checkpoint_dir: str = '/tmp/tacotron/ljspeech/LJSpeech'
    checkpoint_load_run: str = 'train'
    checkpoint_file: Optional[str] = None
    checkpoint_save_run: str = 'inference'
    synthesis_dir: str = '/thesis/inference/ljspeech'
    synthesis_file: str = '/tmp/inference/sentences.txt'
    dump_alignments: bool = True
    dump_linear_spectrogram: bool = True
    n_synthesis_threads: int = 6


class LJSpeechConstants:
    """dB statistics of the LJ-Speech loader (reference datasets/lj_speech.py:20-29)."""
    mel_mag_ref_db = 6.02
    mel_mag_max_db = 99.89
    linear_ref_db = 35.66
    linear_mag_max_db = 100.0


_VOCAB = {
    'pad': 0, 'eos': 1,
    'p': 2, 'r': 3, 'i': 4, 'n': 5, 't': 6, 'g': 7, ' ': 8, 'h': 9, 'e': 10, 'o': 11, 'l': 12,
    'y': 13, 's': 14, 'w': 15, 'c': 16, 'a': 17, 'd': 18, 'f': 19, 'm': 20, 'x': 21, 'b': 22,
    'v': 23, 'u': 24, 'k': 25, 'j': 26, 'z': 27, 'q': 28, ',': 29, '"': 30, '-': 31, ';': 32,
    '(': 33, ')': 34, ':': 35, "'": 36, '!': 37, '?': 38,
}


@dataclass
class DatasetParams:
    dataset_folder: str = '/thesis/datasets/ljspeech'
    dataset_loader: type = LJSpeechConstants  # the text helper lives in datasets.lj_speech
    vocabulary_dict: dict = field(default_factory=lambda: dict(_VOCAB))
    vocabulary_size: int = 39


model_params = ModelParams()
inference_params = InferenceParams()
dataset_params = DatasetParams()
