#!/usr/bin/env python3
"""Golden vectors for VarModel.forward with random_sample = True (spock_reg_model.py:404-408, :502-503): `augment` picks a random number
of timesteps (hparams['samp'] .. T, with replacement, numpy's global generator) behind the column masks and in front of the input noise,
so the network sees a series of another length.  The UNMODIFIED reference, pretrained member v50_0 with the flag set, quiet and noisy,
every draw taped (the two np.random.randint calls included).  Build container only.

    python tests/golden/make_golden_augment.py      # writes case_augment.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Tape, import_reference, pretrained, save  # noqa: E402


def main():
    srm = import_reference()
    torch.set_num_threads(1)
    m = srm.load_swag(pretrained(0)).cpu()
    m.eval()
    m.load(m.w_avg)
    m.random_sample = True
    x = torch.tensor(np.load(os.path.join(HERE, "inputs.npz"))["x_slow"][:16].copy())
    out = {}
    for i, (seed, noisy) in enumerate(((7100, False), (7101, True), (7102, False), (7103, True))):
        np.random.seed(seed)
        torch.manual_seed(seed)
        with Tape() as tape:
            o = m(x, noisy_val=noisy).detach()
        out[f"run{i}_out"] = o.numpy()
        out[f"run{i}_seed"] = np.array(seed)
        out[f"run{i}_noisy"] = np.array(int(noisy))
        out.update(tape.as_dict(f"run{i}_tape"))
        kinds = [k for k, _ in tape.items]
        assert kinds[:2] == ["np.randint", "np.randint"], kinds
        print(i, "T' =", int(tape.items[0][1]), kinds)
    # sample() switches the augmentation off for its loop and restores the flag (:532-543)
    np.random.seed(7200)
    torch.manual_seed(7200)
    with Tape() as tape:
        s = m.sample(x, samples=2)
    assert m.random_sample is True and all(k != "np.randint" for k, _ in tape.items)
    out.update(sample_out=np.asarray(s), **tape.as_dict("sample_tape"))
    save("case_augment.npz", x=x.numpy(), w=m.flatten().detach().numpy(), runs=np.array(4), **out)


if __name__ == "__main__":
    main()
