"""``slender_det.engine`` / ``detectron2.engine`` surface of the hot path (reference: slender_det/engine/defaults.py:22-178,
train_net.py:145-195): argument parser, launcher, default_setup, BaseTrainer."""
from . import hooks
from .defaults import BaseTrainer, DefaultTrainer, default_argument_parser, default_setup
from .launch import launch
