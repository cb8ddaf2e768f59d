"""Text front-end of the inference path: the step immediately before the hot path
(reference datasets/dataset_helper.py, datasets/lj_speech.py)."""
