from .synthetic import SyntheticCocoBatches, synthetic_batch
