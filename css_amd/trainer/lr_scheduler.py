"""Warm-up / fixed / exponential-decay LR schedule.

Drop-in for css/trainer/lr_scheduler.py:8-74 of desh2608/css (Copyright 2020 Matthew Wiesner, Apache 2.0): the CLI flags,
the attribute names and the ``state_dict`` keys below are that file's - they are the checkpoint / command-line surface a
reference ``{e}.mdl`` carries (train.py:304-317) - and the schedule is the same function of the update count
(golden trace: tests/golden/trainer.json).  The schedule itself is stated here as a table of phases.
"""
import math


class LRScheduler(object):
    _FLAGS = (("--warmup", int, 0), ("--decay", float, 0.0), ("--fixed", int, 0), ("--min-lr", float, 1e-09))

    @staticmethod
    def add_args(parser):
        for flag, typ, default in LRScheduler._FLAGS:
            parser.add_argument(flag, type=typ, default=default)

    def __init__(self, optimizer, args):
        self.optimizer = optimizer
        self.warmup, self.fixed, self.decay, self.min_lr = args.warmup, args.fixed, args.decay, args.min_lr
        self.num_warmup_updates = self.num_fixed_updates = self.num_decay_updates = 0
        self.lr = optimizer.param_groups[0]["lr"]
        self.curr_lr = self.lr
        if self.warmup > 0:   # the first updates ramp up from min_lr
            self.curr_lr = self.min_lr
            self.set_lr(self.curr_lr)

    def _phases(self):
        """(length in updates or None = open ended, counter attribute, lr as a function of the counter), in order: a
        phase is active until its counter has reached its length (lr_scheduler.py:33-46)."""
        ramp = (self.lr - self.min_lr) / float(self.warmup) if self.warmup > 0 else 0.0
        return ((self.warmup, "num_warmup_updates", lambda n: self.min_lr + ramp * n),
                (self.fixed, "num_fixed_updates", lambda n: self.lr),
                (None, "num_decay_updates", lambda n: self.lr * math.exp(-self.decay * n)))

    def step(self, num_new_updates):
        for length, counter, rate in self._phases():
            if length is None or (length > 0 and getattr(self, counter) < length):
                n = getattr(self, counter) + num_new_updates
                setattr(self, counter, n)
                self.curr_lr = rate(n)
                break
        self.set_lr(self.curr_lr)

    def set_lr(self, lr):
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    _KEYS = (("warmup", "warmup"), ("fixed", "fixed"), ("decay", "decay"), ("warmup_updates", "num_warmup_updates"),
             ("fixed_updates", "num_fixed_updates"), ("decay_updates", "num_decay_updates"), ("lr", "lr"),
             ("curr_lr", "curr_lr"), ("min_lr", "min_lr"))

    def state_dict(self):
        return {k: getattr(self, a) for k, a in self._KEYS}

    def load_state_dict(self, state_dict):
        for k, a in self._KEYS:
            setattr(self, a, state_dict[k])
