"""safetensors / JSON helpers (reference ``utils/save.py``)."""

import json
from pathlib import Path
from typing import Any, Dict

import torch
from safetensors import safe_open


def load_file(path: Path) -> Dict[str, torch.Tensor]:
    out = {}
    with safe_open(str(path), framework="pt", device="cpu") as handle:
        for key in handle.keys():
            out[key] = handle.get_tensor(key)
    return out


def save_json(obj: Any, path: Path) -> None:
    with open(path, "w", encoding="utf-8") as handle:
        json.dump(obj, handle, indent=4)


def load_json(path: Path) -> Dict[str, Any]:
    with open(path, "r", encoding="utf-8") as handle:
        return json.load(handle)


def verify_models_equivalence(state_dict1: Dict[str, torch.Tensor], state_dict2: Dict[str, torch.Tensor]) -> bool:
    """Whether two state dicts describe the same model: same keys, every tensor equal within ``rtol 1.3e-6 / atol 1e-5`` when
    compared as fp32 on the host (the reference's check that factors are reused with the model they were fitted on,
    utils/save.py:67-101)."""
    if state_dict1.keys() != state_dict2.keys():
        return False
    for name, first in state_dict1.items():
        a, b = first.detach().to(device="cpu", dtype=torch.float32), state_dict2[name].detach().to(device="cpu", dtype=torch.float32)
        if a.shape != b.shape or not torch.allclose(a, b, rtol=1.3e-6, atol=1e-5):
            return False
    return True
