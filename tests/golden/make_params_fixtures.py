#!/usr/bin/env python3
"""Generates tests/golden/params.json from the reference's hyper-parameter singletons.

Runs only in the build container (needs /root/reference).  The three modules build
``tf.contrib.training.HParams(...)`` objects and cannot be imported without TensorFlow; the keyword arguments of those
calls are literals (numbers, strings, tuples, nested HParams calls) plus a few names (``tf.nn.relu``,
``LuongAttention``, ``AttentionScore.DOT``): the calls are read with ``ast`` and stored as data -- names as their dotted
source text.

  tacotron/params/model.py:8-153      model_params
  tacotron/params/inference.py:4-35   inference_params
  tacotron/params/dataset.py:9-32     dataset_params
  datasets/lj_speech.py:20-29         the loader's dB constants (class attributes)
"""
import ast
import json
import os

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def dotted(node):
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Attribute):
        return dotted(node.value) + '.' + node.attr
    raise ValueError(ast.dump(node))


def value(node):
    if isinstance(node, ast.Call):           # a nested HParams(...)
        return {kw.arg: value(kw.value) for kw in node.keywords}
    if isinstance(node, (ast.Tuple, ast.List)):
        return [value(e) for e in node.elts]
    if isinstance(node, ast.Dict):
        return {str(value(k)): value(v) for k, v in zip(node.keys, node.values)}
    if isinstance(node, (ast.Name, ast.Attribute)):
        return {'name': dotted(node)}
    return ast.literal_eval(node)


def hparams_call(path, target):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == target for t in node.targets):
            return value(node.value), [node.lineno, node.end_lineno]
    raise RuntimeError('{}: no assignment to {}'.format(path, target))


def class_attributes(path, cls):
    tree = ast.parse(open(path).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for st in node.body:
                if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
                    try:
                        out[st.targets[0].id] = ast.literal_eval(st.value)
                    except ValueError:
                        pass
            for fn in node.body:            # ... and `self.x = literal` in __init__
                if isinstance(fn, ast.FunctionDef) and fn.name == '__init__':
                    for st in ast.walk(fn):
                        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Attribute) \
                                and isinstance(st.targets[0].value, ast.Name) and st.targets[0].value.id == 'self':
                            try:
                                out[st.targets[0].attr] = ast.literal_eval(st.value)
                            except ValueError:
                                pass
    return out


def main():
    out = {'generated_by': 'tests/golden/make_params_fixtures.py (ast of the reference\'s HParams calls; names kept as source text)'}
    for key, path, target in (('model_params', 'tacotron/params/model.py', 'model_params'),
                              ('inference_params', 'tacotron/params/inference.py', 'inference_params'),
                              ('dataset_params', 'tacotron/params/dataset.py', 'dataset_params')):
        v, lines = hparams_call(os.path.join(REF, path), target)
        out[key] = v
        out[key + '_lines'] = [path] + lines
    out['lj_speech_constants'] = {k: v for k, v in class_attributes(os.path.join(REF, 'datasets/lj_speech.py'), 'LJSpeechDatasetHelper').items()
                                  if isinstance(v, (int, float))}
    with open(os.path.join(HERE, 'params.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:3000])


if __name__ == '__main__':
    main()
