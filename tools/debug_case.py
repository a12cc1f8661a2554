"""Debug helper (GPU): print the worst-error locations of one parity case."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_parity as T

def show(c, backend="mfma", dtype=torch.bfloat16):
    dev = torch.device("cuda:0")
    inp = T.make_inputs(c, dtype)
    ref = T.run_oracle(c, *inp)
    got = T.run_hip(c, *inp, dtype, backend, dev)
    sc = T.run_hip(c, *inp, dtype, "scalar", dev)
    for k in ("out", "dq", "dkv"):
        err = (got[k] - ref[k]).abs()
        v, idx = err.reshape(-1).topk(6)
        print(k, "shape", tuple(err.shape))
        for vi, ii in zip(v.tolist(), idx.tolist()):
            pos = []
            r = ii
            for s in reversed(err.shape):
                pos.append(r % s); r //= s
            pos = tuple(reversed(pos))
            print(f"   err {vi:.4f} at {pos}: got {got[k][pos]:.4f} ref {ref[k][pos]:.4f} scalar {sc[k][pos]:.4f}")

if __name__ == "__main__":
    show(T._case(3, 16, 3, 7, 7, 2))
    show(T._case(3, 32, 7, 96, 96, 1, B=1))
