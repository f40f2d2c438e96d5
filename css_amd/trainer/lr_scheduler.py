"""Warm-up / fixed / exponential-decay LR schedule -- css/trainer/lr_scheduler.py:8-74."""
import math


class LRScheduler(object):
    @staticmethod
    def add_args(parser):
        parser.add_argument("--warmup", type=int, default=0)
        parser.add_argument("--decay", type=float, default=0.0)
        parser.add_argument("--fixed", type=int, default=0)
        parser.add_argument("--min-lr", type=float, default=1e-09)

    def __init__(self, optimizer, args):
        self.optimizer = optimizer
        self.warmup, self.fixed, self.decay, self.min_lr = args.warmup, args.fixed, args.decay, args.min_lr
        self.num_warmup_updates = self.num_fixed_updates = self.num_decay_updates = 0
        self.lr = optimizer.param_groups[0]["lr"]
        self.curr_lr = self.lr
        if self.warmup > 0:
            self.set_lr(self.min_lr)
            self.curr_lr = self.min_lr

    def step(self, num_new_updates):
        if self.warmup > 0 and self.num_warmup_updates < self.warmup:
            self.num_warmup_updates += num_new_updates
            new_lr = self.min_lr + (self.lr - self.min_lr) / float(self.warmup) * self.num_warmup_updates
        elif self.fixed > 0 and self.num_fixed_updates < self.fixed:
            self.num_fixed_updates += num_new_updates
            new_lr = self.lr
        else:
            self.num_decay_updates += num_new_updates
            new_lr = self.lr * math.exp(-self.decay * self.num_decay_updates)
        self.set_lr(new_lr)
        self.curr_lr = new_lr

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
