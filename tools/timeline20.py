import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
sstts = importlib.import_module('single-speaker-tts_amd')
P = importlib.import_module('single-speaker-tts_amd.tacotron.params')
Wm = importlib.import_module('single-speaker-tts_amd.tacotron.weights')
hp = P.ModelParams()
eng = sstts.Engine(hp)
eng.load_weights(Wm.synthetic_weights(0, hp))
ids = eng.to_device(bench.synthetic_ids(64, 150, 1234))
wav = eng.empty((64, 275 * 999))
def step(k):
    eng.synthesize(ids, 200, 6.02, 99.89, 1.3, 60, 1102, 275, seed=k + 1, peak_normalize=True, wav=wav)
for _ in range(5):
    step(0)
eng.set_option('debug_hooks', 1)
eng.set_option('timeline', 1)
eng.set_option('profile', 1)
eng.profile_reset()
eng.synchronize()
t0 = time.perf_counter()
for k in range(20):
    step(k)
eng.synchronize()
print('%.2f ms per step' % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
eng.profile_get('decoder')
