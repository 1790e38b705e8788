"""Which torch ops inside one training step move more than 32 MB (clone / contiguous / copy_ / cat / to / repeat_interleave / zeros / empty.fill_)?
A TorchDispatchMode logs every aten op whose output is larger than that, with the Python frame that called it."""
import sys, traceback
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import bench
from torch.utils._python_dispatch import TorchDispatchMode


MIN_MB = int(__import__("os").environ.get("MIN_MB", "32"))


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for o in outs:
            big = isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() >= (MIN_MB << 20)
            if "cat" in str(func) and isinstance(o, torch.Tensor) and o.is_cuda:
                big = True
            if big and "empty" not in str(func) and "view" not in str(func) \
                    and "as_strided" not in str(func) and "slice" not in str(func) and "select" not in str(func) and "reshape" not in str(func) and "alias" not in str(func) \
                    and "detach" not in str(func) and "unsqueeze" not in str(func) and "permute" not in str(func) and "transpose" not in str(func) and "t.default" not in str(func):
                fr = [f for f in traceback.extract_stack() if "grove_amd" in f.filename or "bench.py" in f.filename][-3:]
                print(f"{str(func):40s} {o.numel() * o.element_size() / 2**20:9.1f} MB {tuple(o.shape)} {o.dtype} <- " + " | ".join(f"{f.filename.split('/')[-1]}:{f.lineno} {f.name}" for f in fr), flush=True)
        return out


class A:
    pass


def main():
    from grove_amd.synthetic import FULL
    args = A()
    args.frames, args.batch, args.text_len, args.exchange, args.stream = 16, 2, 128, "allreduce", "default"
    dev = torch.device("cuda:0")
    model, engine = bench.build(FULL, dev, args)
    batch = bench.make_batch(FULL, dev, args, 0)
    for i in range(2):
        out = engine(**batch); engine.backward(out["loss"]); engine.step()
    torch.cuda.synchronize()
    print("---- logged step")
    with Log():
        out = engine(**batch); engine.backward(out["loss"]); engine.step()
    torch.cuda.synchronize()


main()
