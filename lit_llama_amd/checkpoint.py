"""Streaming checkpoint reader: `lazy_load`.

Replaces the reference's `lazy_load` / `NotYetLoadedTensor` (/root/reference lit_llama/utils.py:166-344), which
leans on private PyTorch reader classes, with a reader of the `torch.save` zip format itself:

    with lazy_load("lit-llama.pth") as checkpoint:        # parses data.pkl only: no tensor bytes are read
        model.load_state_dict(checkpoint)                  # each tensor is mapped when load_state_dict touches it

`torch.save` stores every storage as an uncompressed, 64-byte aligned zip member, so a tensor is a byte range of
the checkpoint file: `LazyTensor.materialize()` maps exactly that range (`numpy.memmap`, zero copy) and views it with
the recorded dtype / size / stride; `.to(device)` then streams it to HBM without ever holding the whole checkpoint
in host memory (LLaMA-65B int4 is 32.5 GB).  Row shards are byte sub-ranges: `narrow(0, start, length)` on a lazy
tensor stays lazy, so a tensor-parallel rank reads only its 1/8 of the column-parallel weights (lit_llama_amd/tp.py,
scripts/convert_checkpoint.py:57-65).

Host-side plumbing only (file parsing, no arithmetic): nothing here runs on the decode path.
"""
from __future__ import annotations

import pickle
import struct
import sys
import warnings
import zipfile
from collections import OrderedDict
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
    "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
    "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}
_SAFE_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "complex",
                  "slice", "range", "bytearray"}


class _StorageRef:
    """One zip member `<archive>/data/<key>`: where its bytes start in the checkpoint file."""

    def __init__(self, reader: "lazy_load", key: str, dtype: torch.dtype, numel: int):
        self.reader, self.key, self.dtype, self.numel = reader, key, dtype, numel


class LazyTensor:
    """A tensor of the checkpoint that has not been read yet: (storage member, element offset, size, stride)."""

    def __init__(self, storage: _StorageRef, offset: int, size: Tuple[int, ...], stride: Tuple[int, ...],
                 requires_grad: bool = False, parameter: bool = False):
        self._storage, self._offset = storage, int(offset)
        self._size, self._stride = tuple(int(s) for s in size), tuple(int(s) for s in stride)
        self._requires_grad, self._parameter = bool(requires_grad), parameter
        self._subclass = None  # (Tensor subclass, its pickled state) when the checkpoint stored one (_rebuild_from_type)

    # ---- metadata without I/O
    @property
    def dtype(self) -> torch.dtype:
        return self._storage.dtype

    @property
    def shape(self) -> torch.Size:
        return torch.Size(self._size)

    @property
    def ndim(self) -> int:
        return len(self._size)

    @property
    def requires_grad(self) -> bool:
        return self._requires_grad

    is_meta = False
    is_sparse = False
    is_quantized = False
    layout = torch.strided
    names = None
    grad = None
    grad_fn = None

    @property
    def device(self) -> torch.device:
        return torch.device("cpu")

    def size(self, dim: Optional[int] = None):
        return self.shape if dim is None else self._size[dim]

    def stride(self, dim: Optional[int] = None):
        return self._stride if dim is None else self._stride[dim]

    def dim(self) -> int:
        return len(self._size)

    def numel(self) -> int:
        n = 1
        for s in self._size:
            n *= s
        return n

    def element_size(self) -> int:
        return torch.empty((), dtype=self.dtype).element_size()

    def __len__(self) -> int:
        return self._size[0]

    def __repr__(self) -> str:
        return f"LazyTensor(shape={tuple(self._size)}, dtype={self.dtype}, member=data/{self._storage.key})"

    # ---- lazy row shards: a sub-range of the same storage member
    def narrow(self, dim: int, start: int, length: int) -> "LazyTensor":
        dim = dim % len(self._size)
        if not (0 <= start and start + length <= self._size[dim]):
            raise IndexError(f"narrow({dim}, {start}, {length}) out of range for {tuple(self._size)}")
        size = list(self._size)
        size[dim] = length
        out = LazyTensor(self._storage, self._offset + start * self._stride[dim], tuple(size), self._stride,
                         self._requires_grad, self._parameter)
        out._subclass = self._subclass
        return out

    # ---- I/O
    def _extent(self) -> Tuple[int, int]:
        """[first, last + 1) element offsets touched inside the storage."""
        if self.numel() == 0:
            return self._offset, self._offset
        last = self._offset + sum((s - 1) * st for s, st in zip(self._size, self._stride))
        return self._offset, last + 1

    def materialize(self) -> torch.Tensor:
        """CPU tensor viewing the mapped byte range of the checkpoint file (read-only memory: copy before writing)."""
        lo, hi = self._extent()
        esz = self.element_size()
        raw = self._storage.reader._map(self._storage.key, lo * esz, (hi - lo) * esz)
        if hi > lo:
            with warnings.catch_warnings():  # the mapping is read-only by design
                warnings.simplefilter("ignore", category=UserWarning)
                flat = torch.from_numpy(raw).view(self.dtype)
        else:
            flat = torch.empty((0,), dtype=self.dtype)
        t = torch.as_strided(flat, self._size, self._stride, 0)
        if self._subclass is not None:  # as the reference does it (lit_llama/utils.py:176-186)
            plain = t
            t = torch._tensor._rebuild_from_type_v2(lambda: plain, self._subclass[0], (), self._subclass[1])
        if self._parameter:
            t = torch.nn.Parameter(t, requires_grad=self._requires_grad)
        return t

    def _load_tensor(self) -> torch.Tensor:
        """The reference's spelling (`NotYetLoadedTensor._load_tensor`, lit_llama/utils.py:166-330): its callers test for
        this attribute and call it (lit_llama/adapter.py:182, scripts/convert_hf_checkpoint.py:65, tests/test_utils.py:49)."""
        return self.materialize()

    def to(self, *args, **kwargs) -> torch.Tensor:
        return self.materialize().to(*args, **kwargs)

    def contiguous(self) -> torch.Tensor:
        return self.materialize().contiguous()

    def __getitem__(self, idx):
        if isinstance(idx, slice) and self._size and idx.step in (None, 1):
            start, stop, _ = idx.indices(self._size[0])
            return self.narrow(0, start, max(0, stop - start))
        return self.materialize()[idx]

    def __getattr__(self, name: str):
        # anything that is not metadata works on the tensor itself (float(), t(), clone(), ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # torch functions / Tensor methods called with a lazy tensor among the arguments (param.copy_(lazy), ...)
        def load(a):
            if isinstance(a, LazyTensor):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(load(x) for x in a)
            return a

        return func(*load(args), **{k: load(v) for k, v in (kwargs or {}).items()})


class _Unpickler(pickle.Unpickler):
    """data.pkl of a torch.save archive, with tensors rebuilt as LazyTensor."""

    def __init__(self, file, reader: "lazy_load"):
        super().__init__(file)
        self.reader = reader

    def find_class(self, module: str, name: str):
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch._tensor" and name == "_rebuild_from_type_v2":
            return _rebuild_from_type
        if module == "torch" and name in _STORAGE_DTYPES:
            return _STORAGE_DTYPES[name]  # the dtype stands in for the legacy storage class
        if module == "torch.storage" and name == "UntypedStorage":
            return torch.uint8
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch" and name == "Size":
            return torch.Size
        if module == "torch" and name in ("device", "dtype"):
            return getattr(torch, name)
        if module == "builtins" and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        # a torch.Tensor SUBCLASS of a module the process has already imported (the reference's tests/test_utils.py:32-49 save
        # one): looked up, never imported — nothing the pickle names gets to run
        # Handed out as an inert marker, NOT as the class: a pickle could otherwise REDUCE-call the class with arguments of its
        # choosing; the marker is only accepted as the `new_type` argument of _rebuild_from_type_v2 (advisor r3)
        obj = getattr(sys.modules.get(module), name, None)
        if isinstance(obj, type) and issubclass(obj, torch.Tensor):
            return _SubclassRef(obj)
        raise pickle.UnpicklingError(f"checkpoint pickle refers to {module}.{name}; only plain state dicts are read")

    def persistent_load(self, pid):
        kind, storage_type, key, _location, numel = pid
        if kind != "storage":
            raise pickle.UnpicklingError(f"unknown persistent id {kind!r}")
        dtype = storage_type if isinstance(storage_type, torch.dtype) else getattr(storage_type, "dtype", torch.uint8)
        return _StorageRef(self.reader, str(key), dtype, int(numel))


def _rebuild_tensor(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    return LazyTensor(storage, storage_offset, tuple(size), tuple(stride), requires_grad)


def _rebuild_parameter(data, requires_grad, backward_hooks):
    if isinstance(data, LazyTensor):
        out = LazyTensor(data._storage, data._offset, data._size, data._stride, requires_grad, parameter=True)
        out._subclass = data._subclass
        return out
    return torch.nn.Parameter(data, requires_grad=requires_grad)


class _SubclassRef:
    """Stands for a torch.Tensor subclass named by a checkpoint pickle; not callable, so the pickle cannot construct it."""

    __slots__ = ("cls",)

    def __init__(self, cls):
        self.cls = cls


def _rebuild_from_type(func, new_type, args, state):
    if isinstance(func, _SubclassRef):
        raise pickle.UnpicklingError("checkpoint pickle calls a Tensor subclass; only plain state dicts are read")
    ret = func(*args)
    if isinstance(ret, LazyTensor) and isinstance(new_type, _SubclassRef):
        ret._subclass = (new_type.cls, state)  # restored at materialize()
    return ret


class lazy_load:
    """Context manager over a `torch.save` checkpoint; yields the unpickled object with LazyTensor leaves.
    `bytes_mapped` counts the bytes handed out so far (what was actually read from the file)."""

    def __init__(self, fn):
        self.path = Path(fn)
        self.bytes_mapped = 0
        self._zf = zipfile.ZipFile(self.path)
        names = self._zf.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            raise ValueError(f"{self.path} is not a torch.save zip archive (data.pkl members: {pkl})")
        self._prefix = pkl[0][: -len("data.pkl")]
        self._data_start: Dict[str, Tuple[int, int]] = {}
        with self._zf.open(pkl[0]) as f:
            self.sd = _Unpickler(f, self).load()

    def _member(self, key: str) -> Tuple[int, int]:
        """(file offset of the member's first data byte, member size)."""
        hit = self._data_start.get(key)
        if hit is None:
            info = self._zf.getinfo(f"{self._prefix}data/{key}")
            if info.compress_type != zipfile.ZIP_STORED:
                raise ValueError(f"{info.filename} is compressed; torch.save stores tensor data uncompressed")
            with open(self.path, "rb") as f:  # the local header's name / extra lengths (extra = alignment padding)
                f.seek(info.header_offset)
                hdr = f.read(30)
            if hdr[:4] != b"PK\x03\x04":
                raise ValueError(f"bad local zip header for {info.filename}")
            n_name, n_extra = struct.unpack("<HH", hdr[26:30])
            hit = (info.header_offset + 30 + n_name + n_extra, info.file_size)
            self._data_start[key] = hit
        return hit

    def _map(self, key: str, byte_off: int, nbytes: int) -> np.ndarray:
        start, size = self._member(key)
        if byte_off < 0 or byte_off + nbytes > size:
            raise ValueError(f"tensor range [{byte_off}, {byte_off + nbytes}) outside storage data/{key} ({size} B)")
        self.bytes_mapped += nbytes
        if nbytes == 0:
            return np.empty((0,), dtype=np.uint8)
        return np.memmap(self.path, dtype=np.uint8, mode="r", offset=start + byte_off, shape=(nbytes,))

    def __enter__(self) -> Any:
        return self.sd

    def __exit__(self, exc_type, exc_val, exc_tb):
        self._zf.close()
        return False
