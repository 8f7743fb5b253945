"""The RCCL code path on real hardware: a process group of ONE rank on the nccl backend (all a single-GPU box allows) runs
the same collectives as N ranks -- device-side all_gather_into_tensor of the per-chunk / per-track logits -- and must give
what the undistributed calls give.  (Two-rank partition / ordering logic: tests/test_parallel_gloo.py on CPU.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from gpu_util import dev

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(180)
def test_sharded_paths_on_a_one_rank_rccl_group_match_the_local_calls():
    from beat_this_amd import weights as W
    from beat_this_amd.inference import Audio2Frames
    from beat_this_amd.model import BeatThis
    from beat_this_amd.parallel import audio2frames_sharded, forward_chunks_sharded

    hp = W.resolve_hparams("small0")
    m = BeatThis(**{k: hp[k] for k in ("spect_dim", "transformer_dim", "ff_mult", "n_layers", "head_dim", "stem_dim")})
    m.load_state_dict(W.random_state_dict(hp, seed=3, style="lively"))
    a2f = Audio2Frames(checkpoint_path=None, device=dev(), float16=False)
    a2f.model = m.to(dev())
    sigs = [W.synthetic_audio(s, seed=80 + i) for i, s in enumerate((31.0, 4.0, 65.0))]
    spects = [torch.from_numpy(W.synthetic_spect(n, seed=90 + i)).to(dev()) for i, n in enumerate((3100, 700, 1500))]

    def frames_fn(sub):
        spect, off = a2f.signal2spect_many(sub, 22050)
        b, d = a2f.spect2frames_batch(spect, off)
        return b, d, off

    local_tracks = audio2frames_sharded(sigs, 22050, frames_fn)
    local_chunks = forward_chunks_sharded(a2f.model, spects)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    try:
        rccl_tracks = audio2frames_sharded(sigs, 22050, frames_fn)
        rccl_chunks = forward_chunks_sharded(a2f.model, spects)
        t = torch.tensor([1.5], dtype=torch.float64, device=dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t) == 1.5
    finally:
        dist.destroy_process_group()
    for (b0, d0), (b1, d1) in zip(local_tracks + local_chunks, rccl_tracks + rccl_chunks):
        assert torch.equal(b0, b1) and torch.equal(d0, d1)
    single = a2f.many(sigs, 22050)
    for (b0, d0), (b1, d1) in zip(single, rccl_tracks):
        assert np.array_equal(b0.cpu().numpy(), b1.cpu().numpy()) and np.array_equal(d0.cpu().numpy(), d1.cpu().numpy())
