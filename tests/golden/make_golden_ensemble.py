#!/usr/bin/env python3
"""Fixtures for the whole 30-member pretrained ensemble (build container only; imports the UNMODIFIED reference through
make_golden.import_reference, reads /root/reference/pretrained/*v50_{0..29}_output.pkl).

  ensemble_v50.npz     w_avg [30,d], w2_avg [30,d], pre_D [30,d,K] float32 in seed-number order (converted states are data),
                       the seed numbers, and per member the count of negative w2_avg - w_avg^2 elements
  case_all_seeds.npz   per member: torch.manual_seed(7000 + i); forward_swag_fast(x_slow[:4], 0.5) by the reference, with the
                       sampled weights and every normal it drew (z1, z2, the two randn_like of compute_summary_stats)

    python tests/golden/make_golden_ensemble.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Tape, import_reference, make_inputs, pretrained, save  # noqa: E402


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    x_slow, _ = make_inputs()
    x = x_slow[:4].contiguous()
    S = 30
    wa, w2, pd, seeds, nneg = [], [], [], [], []
    outs, ws, z1s, z2s, epss = [], [], [], [], []
    for i in range(S):
        m = srm.load_swag(pretrained(i)).cpu()
        m.eval()
        wa.append(m.w_avg.numpy().copy()); w2.append(m.w2_avg.numpy().copy()); pd.append(m.pre_D.numpy().copy())
        seeds.append(int(m.hparams["seed"]))
        nneg.append(int(((m.w2_avg - m.w_avg ** 2) < 0).sum()))
        torch.manual_seed(7000 + i)
        with Tape() as tape:
            out = m.forward_swag_fast(x, 0.5)
        kinds = [k for k, _ in tape.items]
        assert kinds == ["torch.randn", "torch.randn", "torch.randn_like", "torch.randn_like"], kinds
        z1s.append(tape.items[0][1].reshape(-1)); z2s.append(tape.items[1][1].reshape(-1))
        epss.append(np.stack([tape.items[2][1], tape.items[3][1]], 1))      # [B,2,20]
        outs.append(out.detach().numpy().copy()); ws.append(m.flatten().detach().numpy().copy())
        print(f"seed {i}: hparams seed {seeds[-1]}, negative variance elements {nneg[-1]}, mu {outs[-1][:, 0]}", flush=True)
    save("ensemble_v50.npz", w_avg=np.stack(wa), w2_avg=np.stack(w2), pre_D=np.stack(pd), seeds=np.array(seeds),
         negative_variance_elements=np.array(nneg))
    save("case_all_seeds.npz", x=x.numpy(), out=np.stack(outs), w=np.stack(ws), z1=np.stack(z1s), z2=np.stack(z2s),
         eps=np.stack(epss), torch_seed0=np.array(7000), negative_variance_elements=np.array(nneg))


if __name__ == "__main__":
    main()
