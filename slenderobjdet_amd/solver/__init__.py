from .build import FusedSGD, build_lr_scheduler, build_optimizer, get_default_optimizer_params
