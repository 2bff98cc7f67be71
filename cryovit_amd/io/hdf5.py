"""Minimal pure-Python HDF5 reader / writer for the tomogram files on either side of the hot path.

h5py is not installable here (no network) and may be absent on the GPU box, while the on-disk contract of the
reference IS HDF5 (``/root/reference/src/cryovit/run/dino_features.py:109-153`` writes ``data`` and
``labels/<key>`` gzip-compressed and ``dino_features`` contiguous; ``datasets/vit_dataset.py:71-88`` and
``datasets/tomo_dataset.py:89-146`` read them).  When h5py is importable, ``cryovit_amd.io`` uses it; otherwise this
module implements the subset of the HDF5 1.x file format those files use:

  writer  superblock v0, v1 object headers, symbol-table groups (one leaf node per group, <= 32 links),
          contiguous datasets and chunked + deflate datasets (one chunk-index node: <= 64 chunks), little-endian
          u8/i8/u16/i16/u32/i32/u64/i64/f16/f32/f64
  reader  superblock v0-v3, v1 and v2 object headers (incl. continuation blocks), symbol-table groups and compact
          link messages, compact / contiguous / chunked (v1 B-tree, any depth) layouts, deflate + shuffle filters

The test-suite cross-validates both directions against the real libhdf5 (found under /opt/conda in this image) and
against an HDF5 file shipped with scipy.  Format reference: "HDF5 File Format Specification Version 2.0/3.0".
"""

from __future__ import annotations

import struct
import zlib
from pathlib import Path

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
GROUP_LEAF_K, GROUP_INTERNAL_K, CHUNK_K = 16, 16, 32


class H5Error(RuntimeError):
    pass


# =====================================================================================================
# datatype <-> numpy
# =====================================================================================================
_FLOAT_PROPS = {2: (10, 5, 0, 10, 15), 4: (23, 8, 0, 23, 127), 8: (52, 11, 0, 52, 1023)}


def _encode_dtype(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise H5Error("big-endian arrays are not supported by the writer")
    size = dt.itemsize
    if dt.kind in "ui":
        bits0 = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBI", 0x10, bits0, 0, 0, size) + struct.pack("<HH", 0, 8 * size)
    if dt.kind == "f" and size in _FLOAT_PROPS:
        eloc, esize, mloc, msize, bias = _FLOAT_PROPS[size]
        return struct.pack("<BBBBI", 0x11, 0x20, 8 * size - 1, 0, size) + struct.pack(
            "<HHBBBBI", 0, 8 * size, eloc, esize, mloc, msize, bias
        )
    raise H5Error(f"unsupported dtype {dt}")


def _decode_dtype(buf: bytes) -> np.dtype:
    cls_ver, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, 0)
    cls = cls_ver & 0x0F
    order = ">" if (b0 & 1) else "<"
    if cls == 0:
        kind = "i" if (b0 & 0x08) else "u"
        return np.dtype(f"{order}{kind}{size}")
    if cls == 1:
        if size not in _FLOAT_PROPS:
            raise H5Error(f"unsupported float size {size}")
        return np.dtype(f"{order}f{size}")
    raise H5Error(f"unsupported HDF5 datatype class {cls}")


# =====================================================================================================
# writer
# =====================================================================================================
class _WGroup:
    def __init__(self):
        self.children: dict[str, object] = {}


class _WDataset:
    def __init__(self, data: np.ndarray, compression, level: int):
        self.data, self.compression, self.level = data, compression, level


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _ohdr_v1(msgs: list[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


_POOL = None


def _pool():
    """Shared worker threads for chunk (de)compression."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _POOL = ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 2))), thread_name_prefix="h5z")
    return _POOL


class H5Writer:
    """``with H5Writer(path) as f: f.create_dataset("labels/mito", arr, compression="gzip")``"""

    def __init__(self, path):
        self.path = Path(path)
        self.root = _WGroup()
        self._fh = None
        self._pos = 0

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()

    def create_group(self, name: str) -> None:
        self._group_for(name.strip("/").split("/") + [""], create_leaf_parent=True)

    def _group_for(self, parts, create_leaf_parent=False) -> _WGroup:
        g = self.root
        for p in parts[:-1]:
            nxt = g.children.setdefault(p, _WGroup())
            if not isinstance(nxt, _WGroup):
                raise H5Error(f"{p} is a dataset, not a group")
            g = nxt
        return g

    def create_dataset(self, name: str, data, compression=None, compression_opts: int = 4) -> None:
        arr = np.ascontiguousarray(data)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        _encode_dtype(arr.dtype)  # validates
        if compression not in (None, "gzip"):
            raise H5Error("only gzip compression is supported")
        parts = name.strip("/").split("/")
        g = self._group_for(parts)
        if parts[-1] in g.children:
            raise H5Error(f"{name} already exists")
        if len(g.children) >= 2 * GROUP_LEAF_K:
            raise H5Error("too many links in one group for the minimal writer")
        g.children[parts[-1]] = _WDataset(arr, compression, int(compression_opts))

    # ---- serialisation ----------------------------------------------------------------------------------
    def _alloc_write(self, blob: bytes) -> int:
        pad = -self._pos % 8
        if pad:
            self._fh.write(b"\0" * pad)
            self._pos += pad
        addr = self._pos
        self._fh.write(blob)
        self._pos += len(blob)
        return addr

    def _write_dataset(self, ds: _WDataset) -> int:
        arr = ds.data
        rank = arr.ndim
        msgs = [
            _msg(0x0001, struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", d) for d in arr.shape)),
            _msg(0x0003, _encode_dtype(arr.dtype), flags=1),
        ]
        if ds.compression == "gzip" and arr.size > 0 and rank >= 1:
            # chunk along the first axis so that there are at most 64 chunks (one v1 B-tree leaf node)
            c0 = max(1, -(-arr.shape[0] // (2 * CHUNK_K)))
            chunk = (c0,) + tuple(arr.shape[1:])
            nchunk = -(-arr.shape[0] // c0)
            def deflate(i):
                blk = arr[i * c0 : (i + 1) * c0]
                if blk.shape[0] < c0:  # edge chunk: pad to the full chunk extent
                    padded = np.zeros(chunk, dtype=arr.dtype)
                    padded[: blk.shape[0]] = blk
                    blk = padded
                return zlib.compress(blk.tobytes(), ds.level)

            # zlib releases the GIL: chunks are deflated in parallel, written in order (the file layout does not change)
            comps = list(_pool().map(deflate, range(nchunk))) if nchunk > 1 and arr.nbytes >= (1 << 20) else [deflate(i) for i in range(nchunk)]
            entries = [(len(comp), i * c0, self._alloc_write(comp)) for i, comp in enumerate(comps)]
            key_fmt_offsets = rank + 1
            node = bytearray(b"TREE" + struct.pack("<BBHQQ", 1, 0, nchunk, UNDEF, UNDEF))
            for size, off0, addr in entries:
                node += struct.pack("<II", size, 0) + struct.pack("<Q", off0) + b"\0" * (8 * (key_fmt_offsets - 1))
                node += struct.pack("<Q", addr)
            node += struct.pack("<II", 0, 0) + struct.pack("<Q", nchunk * c0) + b"\0" * (8 * (key_fmt_offsets - 1))
            key_size = 8 + 8 * key_fmt_offsets
            full = 24 + (2 * CHUNK_K + 1) * key_size + 2 * CHUNK_K * 8
            node += b"\0" * (full - len(node))
            btree = self._alloc_write(bytes(node))
            msgs.append(_msg(0x0005, struct.pack("<BBBB", 2, 3, 2, 0)))
            msgs.append(_msg(0x000B, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHHI4x", 1, 0, 1, 1, ds.level), flags=1))
            lay = struct.pack("<BBB", 3, 2, rank + 1) + struct.pack("<Q", btree)
            lay += b"".join(struct.pack("<I", c) for c in chunk) + struct.pack("<I", arr.dtype.itemsize)
            msgs.append(_msg(0x0008, lay))
        else:
            addr = self._alloc_write(arr.tobytes()) if arr.nbytes else UNDEF
            msgs.append(_msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)))
            msgs.append(_msg(0x0008, struct.pack("<BB", 3, 1) + struct.pack("<QQ", addr, arr.nbytes)))
        return self._alloc_write(_ohdr_v1(msgs))

    def _write_group(self, g: _WGroup) -> tuple[int, int, int]:
        """returns (object header address, btree address, heap address)"""
        names = sorted(g.children)  # strcmp order
        entries = []
        for n in names:
            c = g.children[n]
            if isinstance(c, _WGroup):
                oh, bt, hp = self._write_group(c)
                entries.append((n, oh, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((n, self._write_dataset(c), 0, b"\0" * 16))
        # local heap: offset 0 = "", then the names; a free block closes the segment
        data = bytearray(b"\0" * 8)
        offs = {}
        for n in names:
            offs[n] = len(data)
            data += _pad8(n.encode() + b"\0")
        free_off = len(data)
        data += struct.pack("<QQ", 1, 32) + b"\0" * 16  # free block: next = 1 (end of list), size 32
        data_addr = self._alloc_write(bytes(data))
        heap = self._alloc_write(b"HEAP" + struct.pack("<B3xQQQ", 0, len(data), free_off, data_addr))
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(entries)))
        for n, oh, ctype, scratch in entries:
            snod += struct.pack("<QQI4x", offs[n], oh, ctype) + scratch
        snod += b"\0" * (8 + 2 * GROUP_LEAF_K * 40 - len(snod))
        snod_addr = self._alloc_write(bytes(snod))
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, UNDEF, UNDEF))
        if entries:
            node += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]])
        node += b"\0" * (24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8 - len(node))
        btree = self._alloc_write(bytes(node))
        ohdr = self._alloc_write(_ohdr_v1([_msg(0x0011, struct.pack("<QQ", btree, heap))]))
        return ohdr, btree, heap

    def close(self) -> None:
        if self._fh is not None:
            return
        with open(self.path, "wb") as fh:
            self._fh = fh
            fh.write(b"\0" * 96)
            self._pos = 96
            root_oh, root_bt, root_hp = self._write_group(self.root)
            eof = self._pos + (-self._pos % 8)
            fh.write(b"\0" * (eof - self._pos))
            sb = SIG + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0)
            sb += struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
            sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
            sb += struct.pack("<QQI4x", 0, root_oh, 1) + struct.pack("<QQ", root_bt, root_hp)
            assert len(sb) == 96
            fh.seek(0)
            fh.write(sb)
        self._fh = True


# =====================================================================================================
# reader
# =====================================================================================================
class H5Dataset:
    def __init__(self, f: "H5Reader", name: str, shape, dtype, layout, filters):
        self._f, self.name, self.shape, self.dtype, self._layout, self._filters = f, name, tuple(shape), dtype, layout, filters

    def __repr__(self):
        return f"<H5Dataset {self.name} {self.shape} {self.dtype}>"

    def read(self) -> np.ndarray:
        kind = self._layout[0]
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        if kind == "compact":
            raw = self._layout[1]
            return np.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape).copy()
        if kind == "contiguous":
            addr, size = self._layout[1], self._layout[2]
            if addr == UNDEF or n == 0:
                return np.zeros(self.shape, dtype=self.dtype)
            return np.fromfile(self._f.path, dtype=self.dtype, count=n, offset=self._f.base + addr).reshape(self.shape)
        if kind == "chunked":
            btree, chunk = self._layout[1], self._layout[2]
            out = np.zeros(self.shape, dtype=self.dtype)
            if btree != UNDEF:
                def inflate(item):
                    mask, offs, raw = item
                    for idx, (fid, _cd) in reversed(list(enumerate(self._filters))):
                        if mask & (1 << idx):  # filter skipped for this chunk
                            continue
                        if fid == 1:
                            raw = zlib.decompress(raw)
                        elif fid == 2:
                            isz = self.dtype.itemsize
                            raw = np.frombuffer(raw, np.uint8).reshape(isz, -1).T.tobytes()
                        else:
                            raise H5Error(f"unsupported filter id {fid}")
                    blk = np.frombuffer(raw, dtype=self.dtype).reshape(chunk)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, self.shape))
                    out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]  # chunks are disjoint: safe from several threads

                # file reads stay serial (one handle); inflating + scattering the chunks runs in parallel (zlib drops the GIL)
                items = [(mask, offs, self._f._read(caddr, csize)) for csize, mask, offs, caddr in self._f._iter_chunks(btree, len(self.shape))]
                if len(items) > 1 and out.nbytes >= (1 << 20):
                    list(_pool().map(inflate, items))
                else:
                    for it in items:
                        inflate(it)
            return out
        raise H5Error(f"unsupported layout {kind}")

    def __getitem__(self, key):
        if key == ():
            return self.read()
        return self.read()[key]


class H5Group:
    def __init__(self, f: "H5Reader", name: str, links: dict[str, int]):
        self._f, self.name, self._links = f, name, links

    def keys(self):
        return list(self._links)

    def __iter__(self):
        return iter(self._links)

    def __contains__(self, k):
        return k.strip("/").split("/")[0] in self._links if k else False

    def __getitem__(self, key: str):
        obj = self
        for p in key.strip("/").split("/"):
            if not isinstance(obj, H5Group) or p not in obj._links:
                raise KeyError(key)
            obj = obj._f._object(obj._links[p], (obj.name.rstrip("/") + "/" + p))
        return obj


class H5Reader(H5Group):
    def __init__(self, path):
        self.path = Path(path)
        self._fh = open(self.path, "rb")
        self.base = self._find_superblock()
        root_addr, root_links = self._parse_superblock()
        links = root_links if root_links is not None else self._group_links(root_addr)
        super().__init__(self, "/", links)

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _read(self, addr: int, n: int) -> bytes:
        self._fh.seek(self.base + addr)
        b = self._fh.read(n)
        if len(b) != n:
            raise H5Error("truncated file")
        return b

    def _find_superblock(self) -> int:
        off = 0
        size = self.path.stat().st_size
        while off < size:
            self._fh.seek(off)
            if self._fh.read(8) == SIG:
                return off
            off = 512 if off == 0 else off * 2  # a user block (e.g. MATLAB v7.3) pushes the superblock to 512 * 2^k
        raise H5Error("not an HDF5 file (no superblock signature)")

    def _parse_superblock(self):
        """Addresses in the file are relative to the base address, which libhdf5 pins to the superblock's own offset."""
        sb = self.base

        def rd(off, n):
            self._fh.seek(sb + off)
            return self._fh.read(n)

        ver = rd(8, 1)[0]
        if ver in (0, 1):
            fixed = 16 if ver == 0 else 20
            hdr = rd(8, fixed)
            if hdr[5] != 8 or hdr[6] != 8:
                raise H5Error("only 8-byte offsets/lengths are supported")
            _name_off, ohdr, _ctype = struct.unpack("<QQI", rd(8 + fixed + 32, 20))
            return ohdr, None
        if ver in (2, 3):
            so, sl, _flags = struct.unpack("<BBB", rd(9, 3))
            if so != 8 or sl != 8:
                raise H5Error("only 8-byte offsets/lengths are supported")
            _base, _ext, _eof, root = struct.unpack("<QQQQ", rd(12, 32))
            return root, None
        raise H5Error(f"unsupported superblock version {ver}")

    # ---- object headers ---------------------------------------------------------------------------------
    def _messages(self, addr: int):
        head = self._read(addr, 4)
        if head == b"OHDR":
            yield from self._messages_v2(addr)
            return
        ver, nmsg, _ref, hsize = struct.unpack("<BxHII", self._read(addr, 12))
        if ver != 1:
            raise H5Error(f"unsupported object header version {ver}")
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            baddr, bsize = blocks.pop(0)
            buf = self._read(baddr, bsize)
            p = 0
            while p + 8 <= bsize and seen < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", buf, p)
                body = buf[p + 8 : p + 8 + msize]
                p += 8 + msize
                seen += 1
                if mtype == 0x0010:
                    caddr, csize = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, csize))
                else:
                    yield mtype, body

    def _messages_v2(self, addr: int):
        ver, flags = struct.unpack("<BB", self._read(addr + 4, 2))
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nbytes = flags & 3
        csize = int.from_bytes(self._read(p, 1 << nbytes), "little")
        p += 1 << nbytes
        blocks = [(p, csize)]
        track = bool(flags & 0x04)
        while blocks:
            baddr, bsize = blocks.pop(0)
            buf = self._read(baddr, bsize)
            q = 0
            while q + 4 <= bsize:
                mtype, msize, mflags = struct.unpack_from("<BHB", buf, q)
                q += 4 + (2 if track else 0)
                body = buf[q : q + msize]
                q += msize
                if mtype == 0x10:
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr + 4, clen - 8))  # skip "OCHK", drop checksum
                elif mtype != 0:
                    yield mtype, body

    def _object(self, addr: int, name: str):
        msgs = list(self._messages(addr))
        types = {t for t, _ in msgs}
        if 0x0008 in types:  # data layout => dataset
            shape, dtype, layout, filters = (), None, None, []
            for t, b in msgs:
                if t == 0x0001:
                    ver, rank, fl = b[0], b[1], b[2]
                    off = 8 if ver == 1 else 4
                    shape = struct.unpack_from(f"<{rank}Q", b, off) if rank else ()
                elif t == 0x0003:
                    dtype = _decode_dtype(b)
                elif t == 0x0008:
                    ver = b[0]
                    if ver == 3:
                        cls = b[1]
                        if cls == 0:
                            (sz,) = struct.unpack_from("<H", b, 2)
                            layout = ("compact", bytes(b[4 : 4 + sz]))
                        elif cls == 1:
                            a, s = struct.unpack_from("<QQ", b, 2)
                            layout = ("contiguous", a, s)
                        elif cls == 2:
                            nd = b[2]
                            (bt,) = struct.unpack_from("<Q", b, 3)
                            dims = struct.unpack_from(f"<{nd}I", b, 11)
                            layout = ("chunked", bt, tuple(dims[:-1]))
                    elif ver in (1, 2):
                        nd, cls = b[1], b[2]
                        p = 8
                        a = UNDEF
                        if cls != 0:
                            (a,) = struct.unpack_from("<Q", b, p)
                            p += 8
                        dims = struct.unpack_from(f"<{nd}I", b, p)
                        layout = ("contiguous", a, 0) if cls == 1 else ("chunked", a, tuple(dims[:-1]))
                    else:
                        raise H5Error(f"unsupported data layout message version {ver} (file written with libver='latest'?)")
                elif t == 0x000B:
                    ver, nf = b[0], b[1]
                    p = 8 if ver == 1 else 2
                    for _ in range(nf):
                        fid, = struct.unpack_from("<H", b, p)
                        if ver == 1 or fid >= 256:
                            nlen, = struct.unpack_from("<H", b, p + 2)
                            fl, ncd = struct.unpack_from("<HH", b, p + 4)
                            p += 8 + (nlen + 7) // 8 * 8 if ver == 1 else 8 + nlen
                        else:
                            fl, ncd = struct.unpack_from("<HH", b, p + 2)
                            p += 6
                        cd = struct.unpack_from(f"<{ncd}I", b, p)
                        p += 4 * ncd + (4 if (ver == 1 and ncd % 2) else 0)
                        filters.append((fid, cd))
            if dtype is None or layout is None:
                raise H5Error(f"{name}: incomplete dataset header")
            return H5Dataset(self, name, shape, dtype, layout, filters)
        return H5Group(self, name, self._group_links(addr, msgs))

    # ---- groups -----------------------------------------------------------------------------------------
    def _group_links(self, addr: int, msgs=None) -> dict[str, int]:
        msgs = msgs if msgs is not None else list(self._messages(addr))
        links: dict[str, int] = {}
        for t, b in msgs:
            if t == 0x0011:
                btree, heap = struct.unpack_from("<QQ", b, 0)
                hsig, _, dsize, _free, daddr = struct.unpack("<4sB3xQQQ", self._read(heap, 32))
                if hsig != b"HEAP":
                    raise H5Error("bad local heap")
                hdata = self._read(daddr, dsize)
                self._walk_group_btree(btree, hdata, links)
            elif t == 0x0006:  # link message (new-style compact group)
                ver, fl = b[0], b[1]
                p = 2
                ltype = 0
                if fl & 0x08:
                    ltype = b[p]
                    p += 1
                if fl & 0x04:
                    p += 8
                if fl & 0x10:
                    p += 1
                lsz = 1 << (fl & 3)
                nlen = int.from_bytes(b[p : p + lsz], "little")
                p += lsz
                nm = b[p : p + nlen].decode()
                p += nlen
                if ltype == 0:
                    links[nm] = struct.unpack_from("<Q", b, p)[0]
            elif t == 0x0002:
                fl = b[1]
                p = 2 + (8 if fl & 1 else 0)
                fheap, _bt = struct.unpack_from("<QQ", b, p)
                if fheap != UNDEF:
                    raise H5Error("dense (fractal-heap) groups are not supported by the minimal reader")
        return links

    def _walk_group_btree(self, addr: int, hdata: bytes, links: dict) -> None:
        sig, ntype, level, used, _l, _r = struct.unpack("<4sBBHQQ", self._read(addr, 24))
        if sig != b"TREE" or ntype != 0:
            raise H5Error("bad group B-tree node")
        body = self._read(addr + 24, (2 * used + 1) * 8)
        for i in range(used):
            child = struct.unpack_from("<Q", body, 8 + 16 * i)[0]
            if level > 0:
                self._walk_group_btree(child, hdata, links)
            else:
                ssig, _v, nsym = struct.unpack("<4sBxH", self._read(child, 8))
                if ssig != b"SNOD":
                    raise H5Error("bad symbol table node")
                ent = self._read(child + 8, nsym * 40)
                for k in range(nsym):
                    noff, oh = struct.unpack_from("<QQ", ent, 40 * k)
                    end = hdata.index(b"\0", noff)
                    links[hdata[noff:end].decode()] = oh

    def _iter_chunks(self, addr: int, rank: int):
        sig, ntype, level, used, _l, _r = struct.unpack("<4sBBHQQ", self._read(addr, 24))
        if sig != b"TREE" or ntype != 1:
            raise H5Error("bad chunk B-tree node")
        ksz = 8 + 8 * (rank + 1)
        body = self._read(addr + 24, used * (ksz + 8) + ksz)
        for i in range(used):
            p = i * (ksz + 8)
            csize, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from(f"<{rank}Q", body, p + 8)
            child = struct.unpack_from("<Q", body, p + ksz)[0]
            if level > 0:
                yield from self._iter_chunks(child, rank)
            else:
                yield csize, mask, offs, child
