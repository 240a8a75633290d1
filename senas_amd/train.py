"""``python -m senas_amd.train --config <yml> [--genotype "Genotype(...)"]``: the train phase (experiments/train_model.py) on the HIP path."""
from .run import train

if __name__ == "__main__":
    train()
