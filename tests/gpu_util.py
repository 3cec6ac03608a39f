"""Helpers for the -m gpu tests: error metrics, a JSON report under gpurun_out/, tensor conversion."""
import json
import os

import numpy as np
import torch

from conftest import ROOT, rel_l2

REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")


def report(name, **vals):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}}) + "\n")


def dev(a, dtype=torch.bfloat16):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(device="cuda", dtype=dtype).contiguous()


def host(t):
    return t.detach().float().cpu()


def bf16r(t):
    """fp32 tensor holding bf16-representable values."""
    return t.to(torch.bfloat16).to(torch.float32)


def errs(got, want):
    g, w = host(got).numpy().astype(np.float64), np.asarray(host(want) if isinstance(want, torch.Tensor) else want, dtype=np.float64)
    return rel_l2(g, w), float(np.abs(g - w).max()), float(np.abs(w).max())
