"""numpy restatement of the pre-path feature packing (TEST INFRASTRUCTURE, like the rest of oracle/).

data_setup_kernel, figures/spock/regression.py:183-213, then StandardScaler.transform (float64) and the cast to
float32 (regression.py:144-145, figures/multiswag_5_planet.py:280-287).  Pinned by tests/golden/case_features.npz,
which make_golden.py produces by executing the reference's own function source."""
import numpy as np

ANGLES = (11, 12, 13, 17, 18, 19, 23, 24, 25)  # regression.py:202


def data_setup(mass_array, cur_tseries):
    """mass_array [3], cur_tseries [1,T,26] -> X [1,T,41] float64."""
    T = cur_tseries.shape[1]
    mass = np.tile(np.asarray(mass_array, np.float64)[None], (T, 1))[None]                    # :185
    old = np.concatenate((np.asarray(cur_tseries, np.float64), mass), axis=2)                  # :187
    for c in (3, 6, 7):                                                                         # :191-193
        old = np.concatenate((old, (~np.isfinite(old[:, :, [c]])).astype(np.float64)), axis=2)
    old = np.nan_to_num(old, posinf=0.0, neginf=0.0)                                            # :195
    cols = []
    for j in range(old.shape[-1]):                                                              # :201-209
        if j in ANGLES:
            cols += [np.cos(old[:, :, [j]]), np.sin(old[:, :, [j]])]
        else:
            cols.append(old[:, :, [j]])
    X = np.concatenate(cols, axis=2)
    if X.shape[-1] != 41:
        raise NotImplementedError("Need to change indexes above for angles, replace ssX.")    # :210-211
    return X


def standardize(X, mean, scale):
    """ssX.transform in float64, then .float()."""
    Xp = (np.asarray(X, np.float64).reshape(-1, X.shape[-1]) - mean) / scale
    return Xp.reshape(X.shape).astype(np.float32)
