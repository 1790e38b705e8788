"""Distribution of the three-step losses / final weights of tests/test_train_gpu.py::test_optimizer_stream_overlap_is_race_free's two arms
(optimizer stream overlapped with the next forward vs not), N repetitions each, interleaved: a race shows as an arm-dependent shift or as
outliers in the overlapped arm; fp32-atomic sum order shows as the same spread in both."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_train_gpu as T  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
out = {True: [], False: []}
for rep in range(N):
    for overlap in (True, False):
        _, args, d, engine = T._engine(dev)
        engine.scheduler.warm = 0
        engine.overlap_optimizer = overlap
        losses = []
        for s in range(3):
            o = engine(**T._batch(d, dev, 10 + s))
            losses.append(o["loss"])
            engine.backward(o["loss"])
            engine.step()
        torch.cuda.synchronize()
        out[overlap].append((torch.stack(losses).float().cpu(), engine.master.clone().cpu()))
for arm in (True, False):
    L = torch.stack([x[0] for x in out[arm]])
    print("overlap" if arm else "serial ", "loss mean", [round(v, 5) for v in L.mean(0).tolist()], "std", [round(v, 5) for v in L.std(0).tolist()],
          "min", [round(v, 4) for v in L.min(0).values.tolist()], "max", [round(v, 4) for v in L.max(0).values.tolist()])
ref = out[False][0][1]
for arm in (True, False):
    print("overlap" if arm else "serial ", "max |w - w(serial run 0)| per run:", [f"{(x[1] - ref).abs().max().item():.2e}" for x in out[arm]])
