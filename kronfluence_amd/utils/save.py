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
