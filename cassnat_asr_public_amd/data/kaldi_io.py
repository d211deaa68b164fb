"""Minimal Kaldi table I/O (binary float/double matrices in .ark files addressed by .scp lines).

The reference delegates this to the third-party `kaldiio` package (src/data/speech_loader.py:6,142, pinned 2.15.1 in
requirements_original.txt), which is not available offline; only the uncompressed binary matrix format that
`copy-feats` / `compute-cmvn-stats` write by default is implemented here.
"""
import mmap
import os
import struct
import threading

import numpy as np

_DTYPES = {b"FM ": np.float32, b"DM ": np.float64}


def _read_matrix(f):
    if f.read(2) != b"\0B":
        raise ValueError("not a binary Kaldi object")
    tag = f.read(3)
    if tag not in _DTYPES:
        raise ValueError("unsupported Kaldi matrix type %r (compressed matrices are not supported)" % tag)
    dims = []
    for _ in range(2):
        if f.read(1) != b"\x04":
            raise ValueError("corrupt Kaldi matrix header")
        dims.append(struct.unpack("<i", f.read(4))[0])
    rows, cols = dims
    dt = np.dtype(_DTYPES[tag]).newbyteorder("<")
    data = np.frombuffer(f.read(rows * cols * dt.itemsize), dtype=dt)
    if data.size != rows * cols:
        raise ValueError("truncated Kaldi matrix")
    return data.reshape(rows, cols).copy()


def load_mat(rxspecifier):
    """"path" or "path:offset" (an .scp entry) -> numpy matrix."""
    path, _, off = rxspecifier.rpartition(":")
    if not path or not off.isdigit():
        path, off = rxspecifier, None
    with open(path, "rb") as f:
        if off is not None:
            f.seek(int(off))
        else:  # a bare file may start with "key "
            head = f.read(2)
            f.seek(0)
            if head != b"\0B":
                while f.read(1) not in (b" ", b""):
                    pass
        return _read_matrix(f)


_maps = {}
_maps_lock = threading.Lock()


def _mapped(path):
    """A read-only memory map of an archive, kept while the file stays the same (a test-set decode reads every matrix of a
    handful of archives once: one map per file instead of one open + seven small reads + a copy per utterance).  The map is
    keyed on the file's identity - inode, size, modification time: an archive rewritten or truncated at the same path gets a new
    map (the stale one stays alive only as long as views into it do)."""
    st = os.stat(path)
    ident = (st.st_ino, st.st_size, st.st_mtime_ns)
    hit = _maps.get(path)
    if hit is None or hit[0] != ident:
        with _maps_lock:
            hit = _maps.get(path)
            if hit is None or hit[0] != ident:
                with open(path, "rb") as f:
                    hit = (ident, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ))
                _maps[path] = hit  # (a replaced map is not closed: numpy views handed out earlier keep it alive)
    return hit[1]


def mat_dtype(rxspecifier):
    """Element type (numpy dtype) of an .scp entry's matrix, from its header."""
    path, _, off = rxspecifier.rpartition(":")
    if not path or not off.isdigit():
        return load_mat(rxspecifier).dtype
    with open(path, "rb") as f:
        f.seek(int(off))
        head = f.read(5)
    if head[:2] != b"\0B" or head[2:5] not in _DTYPES:
        raise ValueError("not a binary Kaldi matrix at %s" % rxspecifier)
    return np.dtype(_DTYPES[head[2:5]])


def load_mat_view(rxspecifier):
    """"path:offset" (an .scp entry) -> a READ-ONLY numpy view of the matrix inside a memory map of the archive (no copy;
    valid as long as the process lives).  Anything else falls back to load_mat."""
    path, _, off = rxspecifier.rpartition(":")
    if not path or not off.isdigit():
        return load_mat(rxspecifier)
    mm = _mapped(path)
    o = int(off)
    head = mm[o : o + 15]
    if head[:2] != b"\0B":
        raise ValueError("not a binary Kaldi object")
    tag = head[2:5]
    if tag not in _DTYPES:
        raise ValueError("unsupported Kaldi matrix type %r (compressed matrices are not supported)" % tag)
    if head[5:6] != b"\x04" or head[10:11] != b"\x04":
        raise ValueError("corrupt Kaldi matrix header")
    rows, cols = struct.unpack("<i", head[6:10])[0], struct.unpack("<i", head[11:15])[0]
    dt = np.dtype(_DTYPES[tag]).newbyteorder("<")
    if o + 15 + rows * cols * dt.itemsize > len(mm):
        raise ValueError("truncated Kaldi matrix")
    return np.frombuffer(mm, dtype=dt, count=rows * cols, offset=o + 15).reshape(rows, cols)


def mat_rows(rxspecifier):
    """Number of rows (frames) of an .scp entry from its 15-byte header alone - what `feat-to-len` gives."""
    path, _, off = rxspecifier.rpartition(":")
    if not path or not off.isdigit():
        return load_mat(rxspecifier).shape[0]
    with open(path, "rb") as f:
        f.seek(int(off))
        head = f.read(10)
    if head[:2] != b"\0B" or head[2:5] not in _DTYPES or head[5:6] != b"\x04":
        raise ValueError("not a binary Kaldi matrix at %s" % rxspecifier)
    return struct.unpack("<i", head[6:10])[0]


def read_scp(scp_path):
    """-> list of (utt, rxspecifier) in file order."""
    out = []
    with open(scp_path, "r") as f:
        for line in f:
            line = line.strip()
            if line:
                utt, spec = line.split(None, 1)
                out.append((utt, spec))
    return out


def write_ark_scp(ark_path, scp_path, items):
    """items: iterable of (utt, matrix).  Writes float32/float64 binary matrices and the matching .scp."""
    with open(ark_path, "wb") as ark, open(scp_path, "w") as scp:
        for utt, mat in items:
            mat = np.ascontiguousarray(mat)
            tag = b"DM " if mat.dtype == np.float64 else b"FM "
            if tag == b"FM ":
                mat = mat.astype("<f4", copy=False)
            ark.write(utt.encode() + b" ")
            scp.write("%s %s:%d\n" % (utt, ark_path, ark.tell()))
            ark.write(b"\0B" + tag + b"\x04" + struct.pack("<i", mat.shape[0]) + b"\x04" + struct.pack("<i", mat.shape[1]))
            ark.write(mat.tobytes())
