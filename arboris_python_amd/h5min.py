"""A minimal HDF5 writer and reader in pure Python (NumPy only), for the observers' output files.

The reference logs simulations with h5py (``arboris/observers.py:133-289``, ``Hdf5Logger``): nested groups of plain
float64 datasets, no attributes, no chunking, no compression -- the files under its ``tests/`` (``simplearm_flat.h5``,
``human36.h5``, ...) look the same.  h5py is not installed next to this package, so ``write`` produces that subset of the
format itself, the way the HDF5 library's own "earliest" file format lays it out (HDF5 File Format Specification,
version 1.1/2.0):

  * superblock version 0, 8-byte offsets and lengths;
  * a group = version-1 object header with a Symbol Table message -> one version-1 B-tree node ("TREE", type 0) -> symbol
    table nodes ("SNOD") whose entries point into the group's local heap ("HEAP") of link names, sorted by name;
  * a dataset = version-1 object header with Dataspace (v1), Datatype (v1: IEEE float / fixed point, little endian),
    Fill Value (v2, default) and Data Layout (v3, contiguous) messages; raw data 8-byte aligned.

``read`` parses the same subset -- and the files the HDF5 library itself wrote for the reference's tests (object header
continuation blocks, contiguous and compact layouts, several symbol table nodes per group) -- into ``{path: ndarray}``.
``tests/test_h5min.py`` reads the reference's own ``human36.h5`` with it, holds ``write -> read`` round trips, and, where
the HDF5 tools are installed (``h5dump``), has the real library read the files written here.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
_INTERNAL_K = 16          # children of a group B-tree node: 2K = 32 symbol table nodes per group


class H5Error(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------------------------------
def _pad8(n):
    return (n + 7) & ~7


def _msg(mtype, data, flags=0):
    data = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages):
    body = b"".join(messages)
    # version 1, reserved, number of messages, reference count 1, size of the message block; padded to 16 bytes
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def _datatype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp_bits, man_bits = (11, 52) if dt.itemsize == 8 else (8, 23)
        bias = (1 << (exp_bits - 1)) - 1
        # class 1 (floating point), version 1; little endian, mantissa normalisation 2 (implied msb), sign bit location
        head = struct.pack("<BBBBI", 0x11, 0x20, dt.itemsize * 8 - 1, 0, dt.itemsize)
        prop = struct.pack("<HHBBBBI", 0, dt.itemsize * 8, man_bits, exp_bits, 0, man_bits, bias)
        return head + prop
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, dt.itemsize * 8)
    raise H5Error("h5min.write: dtype %s is not supported (float32/64, integers)" % dt)


class _Writer(object):
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, nbytes):
        """Reserve `nbytes` at the (8-byte aligned) end of the file; returns the address."""
        self.buf.extend(b"\0" * (_pad8(len(self.buf)) - len(self.buf)))
        addr = len(self.buf)
        self.buf.extend(b"\0" * nbytes)
        return addr

    def put(self, addr, data):
        self.buf[addr:addr + len(data)] = data

    def append(self, data):
        addr = self.alloc(len(data))
        self.put(addr, data)
        return addr

    # -- objects --------------------------------------------------------------------------------------------------
    def dataset(self, arr):
        arr = np.asarray(arr)
        if arr.dtype.kind == "f" and arr.dtype.itemsize not in (4, 8):
            arr = arr.astype(np.float64)
        if arr.dtype.kind == "b":
            arr = arr.astype(np.int8)
        le = arr.dtype.newbyteorder("<") if arr.dtype.byteorder == ">" else arr.dtype
        raw = np.ascontiguousarray(arr, dtype=le).tobytes()
        data_addr = self.append(raw) if raw else UNDEF
        rank = arr.ndim
        space = struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", int(d)) for d in arr.shape)
        fill = struct.pack("<BBBB", 2, 2, 2, 0)          # version 2, late allocation, fill time "if set", no value
        layout = struct.pack("<BBQQ", 3, 1, data_addr, len(raw))
        return self.append(_object_header([_msg(0x0001, space), _msg(0x0003, _datatype_message(le), flags=1),
                                           _msg(0x0005, fill, flags=1), _msg(0x0008, layout)]))

    def group(self, entries, leaf_k):
        """entries: {name: (object header address, (btree, heap) or None)} -> (header, btree, heap) addresses."""
        names = sorted(entries, key=lambda s: s.encode("utf-8"))
        # local heap: offset 0 = the empty string (the left-most B-tree key), then the link names
        heap_data = bytearray(b"\0" * 8)
        offs = {}
        for nm in names:
            b = nm.encode("utf-8") + b"\0"
            offs[nm] = len(heap_data)
            heap_data.extend(b + b"\0" * (_pad8(len(b)) - len(b)))
        # the library wants room for a free block descriptor (16 bytes) at the free-list head
        free_off = len(heap_data)
        heap_data.extend(struct.pack("<QQ", 1, 16))      # next free block: none (1); size of this block
        heap_data_addr = self.append(bytes(heap_data))
        heap_addr = self.append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, heap_data_addr))
        # symbol table nodes of at most 2 * leaf_k entries
        per = 2 * leaf_k
        chunks = [names[i:i + per] for i in range(0, len(names), per)] or [[]]
        if len(chunks) > 2 * _INTERNAL_K:
            raise H5Error("h5min.write: %d links in one group (at most %d)" % (len(names), per * 2 * _INTERNAL_K))
        keys, children = [0], []
        for ch in chunks:
            node = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(ch)))
            for nm in ch:
                oh, grp = entries[nm]
                if grp is None:
                    node.extend(struct.pack("<QQII16x", offs[nm], oh, 0, 0))
                else:
                    node.extend(struct.pack("<QQIIQQ", offs[nm], oh, 1, 0, grp[0], grp[1]))
            node.extend(b"\0" * (8 + per * 40 - len(node)))
            children.append(self.append(bytes(node)))
            keys.append(offs[ch[-1]] if ch else 0)
        tree = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(children) if names else 0, UNDEF, UNDEF))
        for i, c in enumerate(children):
            tree.extend(struct.pack("<QQ", keys[i], c))
        tree.extend(struct.pack("<Q", keys[len(children)]))
        tree.extend(b"\0" * (24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8 - len(tree)))
        btree_addr = self.append(bytes(tree))
        header = self.append(_object_header([_msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]))
        return header, btree_addr, heap_addr


def _tree_of(data):
    root = {}
    for path, v in data.items():
        parts = [p for p in str(path).split("/") if p]
        if not parts:
            raise H5Error("h5min.write: empty dataset path")
        node = root
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise H5Error("h5min.write: %r is both a dataset and a group" % p)
        if isinstance(node.get(parts[-1]), dict):
            raise H5Error("h5min.write: %r is both a dataset and a group" % path)
        node[parts[-1]] = np.asarray(v)
    return root


def write(filename, data):
    """Write ``{path: array}`` (paths with "/" make groups) as an HDF5 file."""
    tree = _tree_of(data)

    def widest(node):
        return max([len(node)] + [widest(v) for v in node.values() if isinstance(v, dict)])
    leaf_k = 4
    while 2 * leaf_k * 2 * _INTERNAL_K < widest(tree):
        leaf_k *= 2
    w = _Writer()
    w.alloc(96)                                       # the superblock, filled in last

    def emit(node):
        entries = {}
        for name, v in node.items():
            if isinstance(v, dict):
                h, b, hp = emit(v)
                entries[name] = (h, (b, hp))
            else:
                entries[name] = (w.dataset(v), None)
        return w.group(entries, leaf_k)
    root_h, root_b, root_hp = emit(tree)
    eof = _pad8(len(w.buf))
    w.buf.extend(b"\0" * (eof - len(w.buf)))
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, _INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQIIQQ", 0, root_h, 1, 0, root_b, root_hp)
    assert len(sb) == 96
    w.put(0, sb)
    with open(filename, "wb") as f:
        f.write(bytes(w.buf))


# ---------------------------------------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------------------------------------
class _Reader(object):
    def __init__(self, raw):
        self.raw = raw
        if raw[:8] != SIGNATURE:
            raise H5Error("not an HDF5 file (signature)")
        ver = raw[8]
        if ver not in (0, 1):
            raise H5Error("h5min.read: superblock version %d (0 and 1 are supported)" % ver)
        if raw[13] != 8 or raw[14] != 8:
            raise H5Error("h5min.read: offsets / lengths of %d / %d bytes (8 / 8 are supported)" % (raw[13], raw[14]))
        pos = 24 if ver == 0 else 28
        self.base, = struct.unpack_from("<Q", raw, pos)
        entry = pos + 32
        self.root_header, = struct.unpack_from("<Q", raw, entry + 8)

    # -- object headers (version 1, with continuation blocks) -------------------------------------------------------
    def messages(self, addr):
        addr += self.base
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", self.raw, addr)
        if ver != 1:
            raise H5Error("h5min.read: object header version %d at %d (1 is supported)" % (ver, addr))
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", self.raw, pos)
                body = self.raw[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                                    # continuation
                    off, length = struct.unpack_from("<QQ", body, 0)
                    blocks.append((off + self.base, length))
                out.append((mtype, flags, body))
        return out

    # -- groups ------------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_addr, off):
        h = heap_addr + self.base
        if self.raw[h:h + 4] != b"HEAP":
            raise H5Error("h5min.read: no local heap at %d" % h)
        data_addr, = struct.unpack_from("<Q", self.raw, h + 24)
        start = data_addr + self.base + off
        end = self.raw.index(b"\0", start)
        return self.raw[start:end].decode("utf-8")

    def _tree_entries(self, node_addr, heap_addr, out):
        a = node_addr + self.base
        sig = self.raw[a:a + 4]
        if sig == b"SNOD":
            n, = struct.unpack_from("<H", self.raw, a + 6)
            for i in range(n):
                noff, oh = struct.unpack_from("<QQ", self.raw, a + 8 + 40 * i)
                out.append((self._heap_name(heap_addr, noff), oh))
            return
        if sig != b"TREE":
            raise H5Error("h5min.read: no group node at %d" % a)
        ntype, level, used = struct.unpack_from("<BBH", self.raw, a + 4)
        if ntype != 0:
            raise H5Error("h5min.read: B-tree node type %d in a group" % ntype)
        for i in range(used):
            child, = struct.unpack_from("<Q", self.raw, a + 24 + 8 + 16 * i)
            self._tree_entries(child, heap_addr, out)

    def links(self, msgs):
        for mtype, _, body in msgs:
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", body, 0)
                out = []
                self._tree_entries(btree, heap, out)
                return out
        return None

    # -- datasets ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(body):
        cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", body, 0)
        cls, order = cv & 0x0F, ">" if (b0 & 1) else "<"
        if cls == 1 and size in (4, 8):
            return np.dtype(order + "f%d" % size)
        if cls == 0 and size in (1, 2, 4, 8):
            return np.dtype(order + ("i" if (b0 & 0x08) else "u") + "%d" % size)
        raise H5Error("h5min.read: datatype class %d of %d bytes is not supported" % (cls, size))

    def dataset(self, msgs, path):
        shape = dtype = layout = None
        for mtype, _, body in msgs:
            if mtype == 0x0001:
                ver, rank, flags = struct.unpack_from("<BBB", body, 0)
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, body, off) if rank else ()
            elif mtype == 0x0003:
                dtype = self._dtype(body)
            elif mtype == 0x0008:
                layout = body
            elif mtype == 0x000B:
                raise H5Error("h5min.read: %s has a filter pipeline (compression): not supported" % path)
        if shape is None or dtype is None or layout is None:
            raise H5Error("h5min.read: %s is neither a group nor a plain dataset" % path)
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[0] != 3:
            raise H5Error("h5min.read: %s: data layout message version %d (3 is supported)" % (path, layout[0]))
        cls = layout[1]
        if cls == 1:                                                   # contiguous
            addr, size = struct.unpack_from("<QQ", layout, 2)
            if addr == UNDEF:
                return np.zeros(shape, dtype.newbyteorder("="))        # never written: the default fill value
            buf = self.raw[addr + self.base:addr + self.base + count * dtype.itemsize]
        elif cls == 0:                                                 # compact
            size, = struct.unpack_from("<H", layout, 2)
            buf = layout[4:4 + size]
        else:
            raise H5Error("h5min.read: %s is chunked: not supported" % path)
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape).astype(dtype.newbyteorder("="))

    def walk(self, addr, prefix, out, seen):
        if addr in seen:
            return
        seen.add(addr)
        msgs = self.messages(addr)
        links = self.links(msgs)
        if links is None:
            out[prefix.rstrip("/") or "/"] = self.dataset(msgs, prefix)
            return
        for name, oh in links:
            self.walk(oh, prefix + name + "/", out, seen)


def read(filename):
    """All datasets of an HDF5 file of the subset described above: ``{"group/name": ndarray}``."""
    with open(filename, "rb") as f:
        raw = f.read()
    r = _Reader(raw)
    out = {}
    r.walk(r.root_header, "", out, set())
    return out
