import sys, time, torch, os
sys.path.insert(0, os.getcwd())
from beat_this_amd import weights as W
from oracle import beat_this_oracle as O
sd = W.random_state_dict('final0', seed=0, style='init')
x = torch.from_numpy(W.synthetic_spect(1500, seed=1))[None]
print('cpu_count', os.cpu_count(), 'default threads', torch.get_num_threads(), flush=True)
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    with torch.inference_mode():
        O.model_forward(sd, x)
        t = time.perf_counter(); O.model_forward(sd, x); dt = time.perf_counter() - t
    print(n, 'threads', round(dt, 3), 's', flush=True)
