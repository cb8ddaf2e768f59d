"""LJ-Speech text helper: dB constants and the abbreviation table of reference
datasets/lj_speech.py (:20-29, :37-60).  Audio loading (``load_audio`` :106-156) is the training
data path and out of scope; the analysis kernels it would use are in ``audio.features``."""
from .dataset_helper import DatasetHelper


class LJSpeechDatasetHelper(DatasetHelper):
    mel_mag_ref_db = 6.02
    mel_mag_max_db = 99.89
    linear_ref_db = 35.66
    linear_mag_max_db = 100.0
    raw_silence_db = None

    def __init__(self, dataset_folder, char_dict, fill_dict):
        super().__init__(dataset_folder, char_dict, fill_dict)
        # order matters: str.replace is applied in insertion order and '.' -> '' must be last
        self._abbreviations = {
            'mr.': 'mister', 'mrs.': 'misses', 'dr.': 'doctor', 'no.': 'number', 'st.': 'saint',
            'co.': 'company', 'jr.': 'junior', 'maj.': 'major', 'gen.': 'general', 'drs.': 'doctors',
            'rev.': 'reverend', 'lt.': 'lieutenant', 'hon.': 'honorable', 'sgt.': 'sergeant',
            'capt.': 'captain', 'esq.': 'esquire', 'ltd.': 'limited', 'col.': 'colonel', 'ft.': 'fort',
            '[': '', ']': '', '.': '',
        }
