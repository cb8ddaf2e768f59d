#!/usr/bin/env python3
"""Generates tests/golden/conversion.npz + conversion.json by RUNNING the reference's own code.

Runs only in the build container (needs /root/reference).  The modules that hold these functions import
librosa / tensorflow at the top and cannot be imported, but the functions themselves are pure numpy / pure
Python: their ``def`` statements are cut out of the reference's source with ``ast`` and executed as they stand,
with numpy as ``np`` -- no stand-in for any library, the functions never touch one:

  audio/conversion.py:5-136      magnitude_to_decibel, decibel_to_magnitude (with its assertion),
                                 normalize_decibel, inv_normalize_decibel, samples_to_ms, ms_to_samples
  tacotron/inference.py:22-27    pad_sentence (reads ``dataset_params.vocabulary_dict['pad']``: the dict literal of
                                 tacotron/params/dataset.py:19-31, parsed as data like make_text_fixtures.py does)

The post-processing chain of tacotron/inference.py:93-101,175 (``inv_normalize_decibel(spectrogram.T, ref, max)`` ->
``decibel_to_magnitude`` -> ``np.power(., magnitude_power)``) is composed here from those extracted functions with
the constants of datasets/lj_speech.py:20-21 and tacotron/params/model.py:45.

Only data leaves this script: inputs and the outputs the reference's functions returned for them.
"""
import ast
import json
import os
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def extract_functions(path, names, namespace):
    """exec the named top-level function definitions of a reference source file in `namespace`."""
    src = open(path).read()
    tree = ast.parse(src)
    found = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, 'exec'), namespace)
            found[node.name] = (node.lineno, node.end_lineno)
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError('{}: no top-level function(s) {}'.format(path, sorted(missing)))
    return found


def dict_literal_after(path, needle):
    src = open(path).read()
    j = src.index('{', src.index(needle))
    depth, k = 0, j
    while True:
        depth += src[k] == '{'
        depth -= src[k] == '}'
        k += 1
        if depth == 0:
            break
    return ast.literal_eval(src[j:k])


def assignment_value(path, name):
    """the literal assigned to `name` (first `name = <literal>` or `name=<literal>` keyword in the file)."""
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets):
            return ast.literal_eval(node.value)
        if isinstance(node, ast.keyword) and node.arg == name:
            return ast.literal_eval(node.value)
    raise RuntimeError('{}: no literal {}'.format(path, name))


def main():
    conv = {'np': np}
    lines = extract_functions(os.path.join(REF, 'audio/conversion.py'),
                              ['magnitude_to_decibel', 'decibel_to_magnitude', 'normalize_decibel', 'inv_normalize_decibel',
                               'samples_to_ms', 'ms_to_samples'], conv)
    vocabulary = dict_literal_after(os.path.join(REF, 'tacotron/params/dataset.py'), 'vocabulary_dict=')
    inf = {'np': np, 'dataset_params': types.SimpleNamespace(vocabulary_dict=vocabulary)}
    lines.update(extract_functions(os.path.join(REF, 'tacotron/inference.py'), ['pad_sentence'], inf))

    ref_db = assignment_value(os.path.join(REF, 'datasets/lj_speech.py'), 'mel_mag_ref_db')
    max_db = assignment_value(os.path.join(REF, 'datasets/lj_speech.py'), 'mel_mag_max_db')
    power = assignment_value(os.path.join(REF, 'tacotron/params/model.py'), 'magnitude_power')
    sr = assignment_value(os.path.join(REF, 'tacotron/params/model.py'), 'sampling_rate')
    win_len_ms = assignment_value(os.path.join(REF, 'tacotron/params/model.py'), 'win_len')
    win_hop_ms = assignment_value(os.path.join(REF, 'tacotron/params/model.py'), 'win_hop')

    rng = np.random.default_rng(20261004)
    arrays = {}

    # --- magnitude_to_decibel: float32 in, both sides of the 1e-5 floor, zero, huge
    mag = np.concatenate([np.float32([0.0, 1e-7, 9.9e-6, 1e-5, 1.0000001e-5, 1e-3, 0.5, 1.0, 2.0, 1234.5, 3e38]),
                          np.exp(rng.uniform(-14, 8, 500)).astype(np.float32)])
    arrays['m2d_in'] = mag
    arrays['m2d_out'] = conv['magnitude_to_decibel'](mag)
    # --- decibel_to_magnitude: float32 in, the whole legal range incl. exactly -100
    db = np.concatenate([np.float32([-100.0, -99.99999, -60.0, -20.0, -6.0206, 0.0, 6.02, 20.0, 35.5]),
                         rng.uniform(-100, 40, 500).astype(np.float32)])
    arrays['d2m_in'] = db
    arrays['d2m_out'] = conv['decibel_to_magnitude'](db)
    # ... and its assertion: anything below -100 dB raises
    asserts = []
    for bad in ([-100.00001], [0.0, -100.5, 3.0], [-1e9]):
        try:
            conv['decibel_to_magnitude'](np.float32(bad))
            asserts.append({'input': bad, 'raises': None})
        except AssertionError as e:
            asserts.append({'input': bad, 'raises': 'AssertionError', 'message': str(e)})
    # --- normalize / inv_normalize with the LJ-Speech constants and with a second pair; both clip sides
    pairs = [(float(ref_db), float(max_db)), (20.0, 100.0), (-3.5, 80.0)]
    for i, (r, m) in enumerate(pairs):
        x = np.concatenate([np.float32([r - abs(r) - abs(m) - 7.0, r - abs(r) - abs(m), r - 1.0, r, r + 0.5, r + 50.0]),
                            rng.uniform(r - abs(r) - abs(m) - 20, r + 20, 300).astype(np.float32)])
        arrays['norm{}_in'.format(i)] = x
        arrays['norm{}_out'.format(i)] = conv['normalize_decibel'](x, r, m)
        y = np.concatenate([np.float32([-0.25, -1e-7, 0.0, 1e-7, 0.5, 1.0 - 1e-7, 1.0, 1.0000001, 1.7]),
                            rng.uniform(-0.3, 1.3, 300).astype(np.float32)])
        arrays['inv{}_in'.format(i)] = y
        arrays['inv{}_out'.format(i)] = conv['inv_normalize_decibel'](y, r, m)
    # --- the chain of tacotron/inference.py:93-101 + 175 on a network-shaped float32 spectrogram (T, F) -> (F, T)
    spec = rng.uniform(-0.2, 1.2, (37, 1025)).astype(np.float32)      # outside [0, 1] too: the clip is part of the chain
    lin_db = conv['inv_normalize_decibel'](spec.T, float(ref_db), float(max_db))
    lin_mag = conv['decibel_to_magnitude'](lin_db)
    arrays['chain_in'] = spec
    arrays['chain_db'] = lin_db
    arrays['chain_mag'] = lin_mag
    arrays['chain_pow'] = np.power(lin_mag, power)

    # --- scalars
    ms = [(50.0, 22050), (12.5, 22050), (25.0, 16000), (8.0, 16000), (12.5, 44100), (0.0, 22050), (1000.0 / 3.0, 48000)]
    scal = {
        'ms_to_samples': [{'ms': a, 'sr': b, 'out': conv['ms_to_samples'](a, b)} for a, b in ms],
        'samples_to_ms': [{'samples': a, 'sr': b, 'out': conv['samples_to_ms'](a, b)} for a, b in [(1102, 22050), (275, 22050), (1, 3)]],
        'model_win': {'win_len_ms': win_len_ms, 'win_hop_ms': win_hop_ms, 'sampling_rate': sr,
                      'win_len': conv['ms_to_samples'](win_len_ms, sr), 'win_hop': conv['ms_to_samples'](win_hop_ms, sr)},
    }
    # --- pad_sentence: int32 ids as process_sentences yields them (np.frombuffer of the bytes), lists, no padding, empty
    pads = []
    for sent, n in ([[5, 6, 7, 1], 9], [[5, 6, 7, 1], 4], [[], 3], [[38, 2, 1], 150]):
        out = inf['pad_sentence'](np.int32(sent), n)
        pads.append({'sentence': sent, 'max_len': n, 'out': out.tolist(), 'dtype': str(out.dtype)})

    out_dtypes = {k: str(v.dtype) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, 'conversion.npz'), **arrays)
    meta = {
        'generated_by': 'tests/golden/make_conversion_fixtures.py (executes the function bodies of the reference)',
        'reference_lines': {k: list(v) for k, v in lines.items()},
        'constants': {'mel_mag_ref_db': ref_db, 'mel_mag_max_db': max_db, 'magnitude_power': power},
        'db_pairs': pairs,
        'dtypes': out_dtypes,
        'decibel_to_magnitude_assertion': asserts,
        'scalars': scal,
        'pad_sentence': pads,
        'pad_token': vocabulary['pad'],
    }
    with open(os.path.join(HERE, 'conversion.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print('wrote conversion.npz ({} arrays) and conversion.json'.format(len(arrays)))
    print('dtypes:', out_dtypes)


if __name__ == '__main__':
    main()
