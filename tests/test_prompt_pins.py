"""Host logic that keeps the pointer-keyed context cache honest (svi_hip.dit.PromptPins, svi_hip.pipeline._stable_bf16).

The C side (svi_dit_context_cache) recognises a prompt embedding by its device pointer.  In the reference's rolling window a new
embedding is made per clip (pipelines/svi_video.py:368-373 encode_prompt per prompt of the stream), so the same address can carry
another prompt a clip later.  These tests run on CPU tensors: the bookkeeping is device-agnostic."""
import gc
import weakref

import torch

from svi_hip.dit import PromptPins, _version
from svi_hip.pipeline import _stable_bf16


def _t(seed, shape=(1, 8, 16)):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(torch.bfloat16)


def test_same_tensor_is_admitted_once():
    pins, a = PromptPins(), _t(0)
    assert pins.admit([a, None]) is False and len(pins) == 1
    assert pins.admit([a]) is False and len(pins) == 1
    assert pins.admit([a[:]]) is False and len(pins) == 1           # a view of the same storage at the same address is the same prompt


def test_pinned_storage_cannot_be_recycled():
    pins, a = PromptPins(), _t(1)
    ref, addr = weakref.ref(a), a.data_ptr()
    pins.admit([a])
    del a
    gc.collect()
    assert ref() is not None                                        # the cache's view of `addr` stays true: the tensor is alive
    later = [_t(10 + i) for i in range(16)]                         # allocations of the same size made afterwards ...
    assert all(t.data_ptr() != addr for t in later)                 # ... never land on the pinned address


def test_in_place_write_drops_the_cache():
    pins, a, b = PromptPins(), _t(2), _t(3)
    assert pins.admit([a, b]) is False
    a.add_(1)
    assert _version(a) > 0
    assert pins.admit([a]) is True                                  # contents behind a cached pointer changed
    assert len(pins) == 1                                           # everything else was forgotten together with the C-side entries
    assert pins.admit([a]) is False


def test_shape_change_at_the_same_address_drops_the_cache():
    pins, a = PromptPins(), _t(4, (1, 8, 16))
    pins.admit([a])
    assert pins.admit([a.view(1, 16, 8)]) is True


def test_capacity_overflow_drops_and_keeps_the_current_call():
    pins = PromptPins(capacity=4)
    keep = [_t(20 + i) for i in range(6)]
    assert [pins.admit([t]) for t in keep[:4]] == [False] * 4
    assert pins.admit([keep[4], keep[5]]) is True                   # 4 + 2 > 4
    assert len(pins) == 2
    assert pins.admit([keep[4], keep[5]]) is False
    old = weakref.ref(keep[0])
    del keep
    gc.collect()
    assert old() is None                                            # dropped pins release their tensors


def test_pair_of_one_call_is_pinned_together():
    """cond / uncond / clip of one CFG step: a drop triggered by one of them must not leave the others unpinned."""
    pins = PromptPins(capacity=3)
    a, b, c, d = _t(30), _t(31), _t(32), _t(33)
    pins.admit([a, b])
    assert pins.admit([c, d, a]) is True
    assert len(pins) == 3 and pins.admit([c, d, a]) is False


def test_inference_tensors_are_accepted():
    with torch.inference_mode():
        a = _t(40)
    assert _version(a) == 0
    assert PromptPins().admit([a]) is False


class _Holder:
    pass


def test_stable_bf16_converts_once_per_tensor():
    h = _Holder()
    a = _t(50)
    assert _stable_bf16(h, a) is a and _stable_bf16(h, None) is None      # already bf16-contiguous: passed through
    f = torch.randn(1, 8, 16)
    c1 = _stable_bf16(h, f)
    assert c1.dtype == torch.bfloat16 and c1.is_contiguous() and torch.equal(c1, f.to(torch.bfloat16))
    assert _stable_bf16(h, f) is c1                                 # same storage, same version: the same copy, the same address
    f.mul_(2)
    c2 = _stable_bf16(h, f)
    assert c2 is not c1 and torch.equal(c2, f.to(torch.bfloat16))   # an in-place write makes a new copy
    nc = _t(51, (1, 16, 8)).transpose(1, 2)                         # bf16 but not contiguous
    c3 = _stable_bf16(h, nc)
    assert c3.is_contiguous() and torch.equal(c3, nc) and _stable_bf16(h, nc) is c3


def test_stable_bf16_keeps_originals_alive_and_bounded():
    h = _Holder()
    f = torch.randn(1, 8, 16)
    ref = weakref.ref(f)
    c = _stable_bf16(h, f)
    del f
    gc.collect()
    assert ref() is not None                                        # its address cannot return as another prompt's while `c` is cached
    for i in range(9):
        _stable_bf16(h, torch.randn(1, 8, 16))
    assert len(h._bf16_memo) <= 8
    del c
