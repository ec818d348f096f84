"""Polynomial decay with linear warm-up.  Mirror of liso/utils/learning_rate.py:4-55 (HF-transformers schedule)."""
from torch.optim.lr_scheduler import LambdaLR


def get_polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, lr_end=1e-7, power=1.0,
                                              last_epoch=-1):
    lr_init = optimizer.defaults["lr"]
    if not (lr_init >= lr_end):
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

    def lr_lambda(step: int):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        if step > num_training_steps:
            return lr_end / lr_init
        remaining = 1 - (step - num_warmup_steps) / (num_training_steps - num_warmup_steps)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init

    return LambdaLR(optimizer, lr_lambda, last_epoch)
