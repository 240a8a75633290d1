"""Model factory with the reference's call shape (models/__init__.py:26-31,92-105):
``get_segmentation_model('senas', dataset=..., c=..., depth=..., supervision=..., genotype=...,
double_down_channel=...)``.  The reference looks ``NUM_CLASS`` / ``IN_CHANNELS`` up on the dataset
class; synthetic benchmarks have no dataset class, so ``nclass=`` / ``in_channels=`` may be given
directly (they win over the dataset lookup)."""
from .senas_model import SenasModel

# NUM_CLASS, IN_CHANNELS of the 2-d datasets the senas configs name (utils/datasets/*.py class attributes)
DATASET_SHAPES = {'promise12': (2, 1), 'chaos': (2, 1), 'monusac': (2, 1), 'heart': (2, 1), 'spleen': (2, 1),
                  'hippo': (3, 1), 'pancreas': (3, 1)}


def senas(dataset='promise12', nclass=None, in_channels=None, **kwargs):
    if nclass is None or in_channels is None:
        d_nclass, d_in = DATASET_SHAPES[str(dataset).lower()]
        nclass = d_nclass if nclass is None else nclass
        in_channels = d_in if in_channels is None else in_channels
    return SenasModel(nclass, in_channels, **kwargs)


def get_segmentation_model(name, **kwargs):
    models = {'senas': senas}
    return models[name.lower()](**kwargs)
