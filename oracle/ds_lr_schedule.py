"""TEST INFRASTRUCTURE ONLY (imported by tests/): restatement of DeepSpeed's `WarmupLR` / `WarmupDecayLR` learning-rate schedules
(deepspeed/runtime/lr_schedules.py at the reference's pin, deepspeed==0.15.1 — /root/reference/requirements.txt:6) and of the order
in which `DeepSpeedEngine` drives them (`_take_model_step`: optimizer.step() first, then lr_scheduler.step()).

PARITY UNPINNED: DeepSpeed is a third-party dependency that is absent from /root/reference and not installed offline, so this
is the published algorithm written down from its source, not something checked against a run of it; no golden lr trace exists.
What it pins is grove_amd.train.WarmupDecayLR.for_update's INDEX arithmetic against a step-by-step simulation of the calling order
the reference uses (train.py:466-486: `deepspeed.initialize(config=ds_config)` builds the scheduler with last_batch_iteration = -1)."""
import math


class _ParamGroups:
    """Stand-in for a torch optimizer: only `param_groups[i]['lr']` is touched by the schedules."""

    def __init__(self, lr, n=1):
        self.param_groups = [{"lr": lr} for _ in range(n)]


def _update_lr(param_groups, lrs):
    for g, lr in zip(param_groups, lrs):
        g["lr"] = lr
    return [g["lr"] for g in param_groups]


class WarmupLR:
    def __init__(self, optimizer, warmup_min_lr=0.0, warmup_max_lr=0.001, warmup_num_steps=1000, warmup_type="log", last_batch_iteration=-1):
        self.optimizer = optimizer
        n = len(optimizer.param_groups)
        self.min_lrs, self.max_lrs = [warmup_min_lr] * n, [warmup_max_lr] * n
        self.delta_lrs = [b - s for b, s in zip(self.max_lrs, self.min_lrs)]
        self.warmup_num_steps = max(2, warmup_num_steps)
        assert warmup_type in ("log", "linear")
        self.warmup_type = warmup_type
        self.inverse_log_warm_up = 1.0 / math.log(self.warmup_num_steps)
        self.last_batch_iteration = last_batch_iteration
        if last_batch_iteration == -1:  # "Initialize lr in optimizer": get_lr() before the first step returns min_lrs
            self._last_lr = _update_lr(self.optimizer.param_groups, self.get_lr())

    def get_lr(self):
        if self.last_batch_iteration < 0:
            return list(self.min_lrs)
        gamma = self._get_gamma()
        return [m + d * gamma for m, d in zip(self.min_lrs, self.delta_lrs)]

    def step(self, last_batch_iteration=None):
        if last_batch_iteration is None:
            last_batch_iteration = self.last_batch_iteration + 1
        self.last_batch_iteration = last_batch_iteration
        self._last_lr = _update_lr(self.optimizer.param_groups, self.get_lr())

    def _warm_gamma(self):
        if self.warmup_type == "log":
            return self.inverse_log_warm_up * math.log(self.last_batch_iteration + 1)
        return self.last_batch_iteration / self.warmup_num_steps

    def _get_gamma(self):
        return self._warm_gamma() if self.last_batch_iteration < self.warmup_num_steps else 1.0


class WarmupDecayLR(WarmupLR):
    def __init__(self, optimizer, total_num_steps, **kw):
        self.total_num_steps = total_num_steps
        super().__init__(optimizer, **kw)

    def _get_gamma(self):
        if self.last_batch_iteration < self.warmup_num_steps:
            return self._warm_gamma()
        return max(0.0, float(self.total_num_steps - self.last_batch_iteration) / float(max(1.0, self.total_num_steps - self.warmup_num_steps)))


def lr_of_updates(base_lr, total_num_steps, n_updates, warmup_num_steps=100):
    """The lr each of the first n optimizer updates runs with under the reference's config (train.py:471-474) and the engine's order."""
    opt = _ParamGroups(base_lr)  # AdamW(lr=args.lr): overwritten by the scheduler's constructor
    sched = WarmupDecayLR(opt, total_num_steps, warmup_min_lr=0, warmup_max_lr=base_lr, warmup_num_steps=warmup_num_steps, warmup_type="linear")
    used = []
    for _ in range(n_updates):
        used.append(opt.param_groups[0]["lr"])  # optimizer.step()
        sched.step()                            # then lr_scheduler.step()
    return used
