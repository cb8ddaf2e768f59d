"""Text -> id-sequence front-end (host logic; mirrors reference datasets/dataset_helper.py).

Only what ``tacotron/inference.py`` calls is reproduced: ``process_sentences`` (:146-241) with its
helpers ``sent2idx`` (:51-68), ``idx2sent`` (:70-87), ``replace_abbreviations`` (:127-144),
``utf8_to_ascii`` (:89-109), ``update_char_dict`` (:111-125) and the static
``apply_reduction_padding`` (:357-401).  Corpus loading / feature pre-computation are out of scope.

Quirks kept on purpose: abbreviations are applied with ``str.replace`` in dict order (so
``'.' -> ''`` must come last), characters outside the vocabulary raise ``KeyError``, the EOS id is
appended, and every id sequence is returned as the raw bytes of an int32 array."""
import numpy as np


class DatasetHelper(object):
    def __init__(self, dataset_folder, char_dict, fill_dict):
        self._dataset_folder = dataset_folder
        self._char2idx_dict = char_dict
        self._fill_dict = fill_dict
        self._abbreviations = dict()
        self._statistics = dict()
        self._idx2char_dict = {_id: char for char, _id in self._char2idx_dict.items()}

    def sent2idx(self, sentence):
        return [self._char2idx_dict[char] for char in sentence]

    def idx2sent(self, idx):
        return ''.join([self._idx2char_dict[_id] for _id in idx])

    def utf8_to_ascii(self, sentence):
        return bytes(sentence, 'utf-8').decode('ascii', errors='ignore')

    def update_char_dict(self, sentence):
        for char in sentence:
            if char not in self._char2idx_dict:
                _id = len(self._char2idx_dict)
                self._char2idx_dict[char] = _id
                self._idx2char_dict[_id] = char

    def replace_abbreviations(self, sentence):
        for abbreviation, expansion in self._abbreviations.items():
            sentence = sentence.replace(abbreviation, expansion)
        return sentence

    def get_statistics(self):
        return self._statistics

    def process_sentences(self, sentences):
        """-> (id_sequences: list of bytes (int32 arrays), sequence_lengths incl. EOS)."""
        word_set, character_set = set(), set()
        st = self._statistics
        st['n_words_total'] = st['n_chars_total'] = 0
        for sentence in sentences:
            sentence = sentence.lower()
            words = sentence.split(' ')
            st['n_words_total'] += len(words)
            word_set.update(words)
            st['n_chars_total'] += len(sentence)
            character_set.update(list(sentence))
        st['n_words_unique'] = len(word_set)
        st['n_chars_unique'] = len(character_set)
        st['n_words_clip_avg'] = st['n_words_total'] / len(sentences)
        st['n_chars_clip_avg'] = st['n_chars_total'] / len(sentences)

        eos_token = self._char2idx_dict['eos']
        id_sequences, sequence_lengths = [], []
        for sentence in sentences:
            sentence = self.replace_abbreviations(sentence.lower())
            if self._fill_dict:
                self.update_char_dict(sentence)
            idx = self.sent2idx(sentence)
            idx.append(eos_token)
            id_sequences.append(np.array(idx, dtype=np.int32).tobytes())
            sequence_lengths.append(len(idx))
        return id_sequences, sequence_lengths

    @staticmethod
    def apply_reduction_padding(mel_mag_db, linear_mag_db, reduction_factor):
        """Zero-pad the frame axis to a multiple of r and fold r frames into one (:357-401)."""
        n_frames = mel_mag_db.shape[0]
        if n_frames % reduction_factor != 0:
            pad = reduction_factor - (n_frames % reduction_factor)
            mel_mag_db = np.pad(mel_mag_db, [[0, pad], [0, 0]], mode='constant')
            linear_mag_db = np.pad(linear_mag_db, [[0, pad], [0, 0]], mode='constant')
        mel_mag_db = mel_mag_db.reshape((-1, mel_mag_db.shape[1] * reduction_factor))
        linear_mag_db = linear_mag_db.reshape((-1, linear_mag_db.shape[1] * reduction_factor))
        return mel_mag_db, linear_mag_db
