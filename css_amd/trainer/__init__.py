from .lr_scheduler import LRScheduler  # noqa: F401
from .iteration_types import train_one_epoch, validate  # noqa: F401
from .graphed import GraphedTrainStep  # noqa: F401
