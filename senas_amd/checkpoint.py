"""Checkpoint interchange with the reference's drivers.

Same file names and dictionary keys as ``utils/utils.py:138-143`` (``save_checkpoint``),
``experiments/search_arc.py:150-175,227-238`` (search phase) and ``experiments/train_model.py:150-175,221-230``
(train phase): a checkpoint written by the reference loads here and vice versa, because the modules of this
package keep the reference's ``state_dict`` keys, shapes and orders, weights stay in the torch layouts
(``Conv2d [co][ci][kh][kw]``, ``ConvTranspose2d [ci][co][kh][kw]`` -- the NHWC/MFMA images are derived data,
rebuilt on the device every step) and the optimizer state is the torch optimizers' own.
"""
import os
import shutil

import torch

CHECKPOINT_NAME = 'checkpint.pth.tar'          # sic: utils/utils.py:139
BEST_NAME = 'model_best.pth.tar'


def save_checkpoint(state, is_best, save):
    """utils/utils.py:138-143."""
    filename = os.path.join(save, CHECKPOINT_NAME)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, os.path.join(save, BEST_NAME))
    return filename


def search_state(model, arch_optimizer, model_optimizer, scheduler, epoch, dur_time=0.0, patience=0, geno_type=''):
    """The dictionary search_arc.py:227-238 saves after every epoch."""
    return {
        'epoch': epoch + 1,
        'dur_time': dur_time,
        'cur_patience': patience,
        'geno_type': geno_type,
        'model_state': model.state_dict(),
        'arch_optimizer': arch_optimizer.state_dict(),
        'model_optimizer': model_optimizer.state_dict(),
        'alphas_dict': model.alphas_dict(),
        'betas_dict': model.betas_dict(),
        'scheduler': scheduler.state_dict() if scheduler is not None else None,
    }


def load_search_state(checkpoint, model, arch_optimizer=None, model_optimizer=None, scheduler=None):
    """search_arc.py:150-175 (``_check_resume``).  Returns (start_epoch, dur_time, geno_type).

    The reference's ``load_params`` reads keys its ``alphas_dict`` never writes (senas_search.py:170-198) and
    then re-binds the architecture tensors, detaching them from the optimizer; here the values are copied into
    the existing parameters instead, so the optimizer state loaded above keeps pointing at live tensors."""
    model.load_state_dict(checkpoint['model_state'])
    if arch_optimizer is not None and checkpoint.get('arch_optimizer') is not None:
        arch_optimizer.load_state_dict(checkpoint['arch_optimizer'])
    if model_optimizer is not None and checkpoint.get('model_optimizer') is not None:
        model_optimizer.load_state_dict(checkpoint['model_optimizer'])
    if scheduler is not None and checkpoint.get('scheduler') is not None:
        scheduler.load_state_dict(checkpoint['scheduler'])
    legacy = {'alphas_down': 'alphas_dn', 'alphas_normal_down': 'alphas_dn_nm', 'alphas_normal_up': 'alphas_up_nm',
              'betas_down': 'betas_dn'}
    with torch.no_grad():
        for d in (checkpoint.get('alphas_dict') or {}, checkpoint.get('betas_dict') or {}):
            for key, value in d.items():
                getattr(model, legacy.get(key, key)).copy_(value)
    return checkpoint.get('epoch', 0), checkpoint.get('dur_time', 0.0), checkpoint.get('geno_type', '')


def train_state(model, model_optimizer, epoch, dur_time=0.0, best_pixAcc=0.0, best_mIoU=0.0, best_dice_coeff=0.0,
                best_loss=float('inf')):
    """The dictionary train_model.py:221-230 saves when the validation loss improves."""
    return {
        'epoch': epoch + 1,
        'dur_time': dur_time,
        'model_state': model.state_dict(),
        'model_optimizer': model_optimizer.state_dict(),
        'best_pixAcc': best_pixAcc,
        'best_mIoU': best_mIoU,
        'best_dice_coeff': best_dice_coeff,
        'best_loss': best_loss,
    }


def load_train_state(checkpoint, model, model_optimizer=None):
    """train_model.py:150-175.  Accepts a bare ``state_dict`` too (testing_model.py loads ``model_state`` only)."""
    state = checkpoint['model_state'] if 'model_state' in checkpoint else checkpoint
    model.load_state_dict(state)
    if model_optimizer is not None and isinstance(checkpoint, dict) and checkpoint.get('model_optimizer') is not None:
        model_optimizer.load_state_dict(checkpoint['model_optimizer'])
    return checkpoint.get('epoch', 0) if 'model_state' in checkpoint else 0
