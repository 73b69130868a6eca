"""Device-memory plumbing for the GPU tier (torch is used for allocation/copies only)."""
import numpy as np


def to_device(arr: np.ndarray, misalign: int = 0):
    import torch
    arr = np.ascontiguousarray(arr)
    if misalign:
        t = torch.empty(arr.size + misalign, dtype=torch.uint8, device="cuda")
        view = t[misalign:]
        view.copy_(torch.from_numpy(arr.view(np.uint8).reshape(-1)))
        return view
    return torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).cuda()


def from_device(t) -> np.ndarray:
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy().copy()
