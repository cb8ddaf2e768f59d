#!/usr/bin/env python3
"""Generates tests/golden/text_frontend.json by RUNNING the reference's own text front-end.

Runs only in the build container (needs /root/reference).  It imports the reference's
``datasets.dataset_helper.DatasetHelper`` (numpy only) and feeds it the LJ-Speech abbreviation
table and vocabulary parsed -- as data -- out of the reference's source files
(datasets/lj_speech.py:37-60, tacotron/params/dataset.py:19-31; both modules import
librosa / tensorflow at the top and cannot be imported themselves).
"""
import ast
import json
import os
import sys

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text_frontend.json')


def dict_literal_after(path, needle):
    src = open(path).read()
    i = src.index(needle)
    j = src.index('{', i)
    depth, k = 0, j
    while True:
        depth += src[k] == '{'
        depth -= src[k] == '}'
        k += 1
        if depth == 0:
            break
    return ast.literal_eval(src[j:k])


def main():
    sys.path.insert(0, REF)
    from datasets.dataset_helper import DatasetHelper   # the reference's class

    abbreviations = dict_literal_after(os.path.join(REF, 'datasets/lj_speech.py'), 'self._abbreviations =')
    vocabulary = dict_literal_after(os.path.join(REF, 'tacotron/params/dataset.py'), 'vocabulary_dict=')

    class RefLJ(DatasetHelper):
        def __init__(self):
            super().__init__('/nonexistent', dict(vocabulary), False)
            self._abbreviations = abbreviations

        def load(self, *a, **k):
            raise NotImplementedError

        def load_audio(self, *a, **k):
            raise NotImplementedError

    ds = RefLJ()
    sentences = [
        'Tis a test!',
        'Printing, in the only sense with which we are at present concerned, differs '
        'from most if not from all the arts and crafts represented in the Exhibition',
        'Neild gives, on the authority of Mr. Burchell, the under sheriff of Middlesex,',
        'Dr. Smith and Mrs. Jones met Capt. Hook at St. James Co. Ltd. [sic].',
        'No. five is alive; "quoted" (text) - done: yes? no!',
        "It's the colonel's fort, Col. Ft. Lt. Gen. Maj. Jr. Hon. Sgt. Rev. Esq. Drs.",
        'a',
        'HELLO WORLD',
    ]
    ids, lens = ds.process_sentences(sentences)
    folded = [ds.replace_abbreviations(s.lower()) for s in sentences]
    mel = np.arange(7 * 3, dtype=np.float32).reshape(7, 3)
    lin = np.arange(7 * 5, dtype=np.float32).reshape(7, 5)
    rmel, rlin = DatasetHelper.apply_reduction_padding(mel, lin, 5)
    out = {
        'vocabulary': vocabulary,
        'abbreviations': list(abbreviations.items()),
        'sentences': sentences,
        'folded': folded,
        'ids': [np.frombuffer(b, dtype=np.int32).tolist() for b in ids],
        'lengths': [int(x) for x in lens],
        'reduction_padding': {'mel_shape': list(rmel.shape), 'lin_shape': list(rlin.shape),
                              'mel': rmel.tolist(), 'lin': rlin.tolist()},
        'known_error': None,
    }
    try:
        ds.process_sentences(['café #1'])
    except KeyError as e:
        out['known_error'] = {'sentence': 'café #1', 'raises': 'KeyError', 'key': e.args[0]}
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1, ensure_ascii=True)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
