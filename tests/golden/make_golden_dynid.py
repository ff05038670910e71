"""tests/golden/dynid.npz: loss_func of model/Dynamic_parallel_model.py:30-42 (AST-extracted from the reference tree where it
lies, executed here once; only inputs and outputs are stored) on seeded random recordings.
    python tests/golden/make_golden_dynid.py      (build container only: needs /root/reference)"""
import os

import numpy as np

from make_golden import REF, OUT, extract


def main():
    ns = extract(REF + "model/Dynamic_parallel_model.py", {"loss_func"})
    rng = np.random.default_rng(20260927)
    T, n = 100, 6
    mean_dict = {}
    for key in ("exp", "ori"):
        mean_dict[key + "_motor_mean"] = rng.normal(size=(T, 12)) * 0.3
        mean_dict[key + "_motor_std"] = rng.uniform(0.05, 0.3, size=(T, 12))
        mean_dict[key + "_drpy_mean"] = rng.normal(size=(T, 3))
        mean_dict[key + "_drpy_std"] = rng.uniform(0.2, 1.0, size=(T, 3))
    motor = rng.normal(size=(n, T, 12)) * 0.4
    drpy = rng.normal(size=(n, T, 3)) * 1.5
    loss = {key: np.array([ns["loss_func"](drpy[i], motor[i], mean_dict, key) for i in range(n)]) for key in ("exp", "ori")}
    np.savez(os.path.join(OUT, "dynid.npz"), motor=motor, drpy=drpy, loss_exp=loss["exp"], loss_ori=loss["ori"], **mean_dict)
    print("dynid.npz written", loss)


if __name__ == "__main__":
    main()
