"""Checkpoint I/O with the reference's file layout.

Mirrors utilities.save_model / save_checkpoint / load_checkpoint / latest_file (utilities.py:162-204): a checkpoint is
`torch.save({'epoch', 'model_state_dict', 'optimizer_state_dict', 'loss'}, f"{dir}/{name}.pt")`, a bare model file is
`torch.save(model.state_dict(), path)`.  The drop-in model has the reference's state_dict keys and FusedAdam speaks
torch.optim.Adam's state layout, so files written by either side load on the other.  Files are read with
`torch.load(..., weights_only=True)`: nothing from the file is executed.
"""
import glob
import os

import torch


def save_model(path, model, overwrite=False):
    torch.save(model.state_dict(), path)


def save_checkpoint(model_dict, opt_dict, epoch, loss, name, dir):
    print(f"[maavss_amd] writing checkpoint {dir}/{name}.pt (epoch {epoch}, validation loss {loss})")
    torch.save({"epoch": epoch, "model_state_dict": model_dict, "optimizer_state_dict": opt_dict, "loss": loss},
               f"{dir}/{name}.pt")


def latest_file(dir, ext):
    all_files = glob.glob(f"{dir}/*.{ext}", recursive=True)
    return max(all_files, key=os.path.getctime) if all_files else None


def load_checkpoint(model, optimizer, dir, auto=True, path=None, load_opt=False):
    """Same contract as the reference: newest *.pt of `dir` (auto) or `path`; model weights with strict=False; the
    optimizer state only on request, and a failure there is reported, not raised.  Returns the checkpoint dict
    (the reference returns None; callers that ignore the result are unaffected)."""
    if auto:
        path = latest_file(dir, "pt")
        if path is None:
            print(f"[maavss_amd] no *.pt checkpoint in {dir}: nothing loaded")
            return None
    elif path is None:
        return None
    print(f"[maavss_amd] reading checkpoint {path}")
    map_location = next(model.parameters()).device
    checkpoint = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(checkpoint["model_state_dict"], strict=False)
    if load_opt:
        try:
            optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        except Exception as e:      # noqa: BLE001 -- reference behaviour (utilities.py:193-196)
            print(f"[maavss_amd] optimizer state of {path} not restored: {e}")
    return checkpoint
