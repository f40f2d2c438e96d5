"""Oracle: LR schedule (reference css/trainer/lr_scheduler.py:33-46).  Test infrastructure only."""
import math


def lr_trace(lr, n_steps, warmup=0, fixed=0, decay=0.0, min_lr=1e-9):
    """LR seen at the start of each of n_steps updates (curr_lr before each step)."""
    cur = min_lr if warmup > 0 else lr
    n_w = n_f = n_d = 0
    out = []
    for _ in range(n_steps):
        out.append(cur)
        if warmup > 0 and n_w < warmup:
            n_w += 1
            cur = min_lr + (lr - min_lr) / float(warmup) * n_w
        elif fixed > 0 and n_f < fixed:
            n_f += 1
            cur = lr
        else:
            n_d += 1
            cur = lr * math.exp(-decay * n_d)
    return out
