"""Minimal HDF5 reader / writer for Keras weight files (no h5py in this environment).

The reference keeps every checkpoint as Keras HDF5 weights and loads them BY LAYER NAME
(dense_img_cap/dense_model.py:1656-1692 load_weights -> keras.engine.topology.load_weights_from_hdf5_group_by_name;
saved by ModelCheckpoint(save_weights_only=True), text_generation_model.py:418,461; _v2.py:256,303): that is how
mask_rcnn_coco.h5 / rcnn_coco.h5 seed the backbone + RoI head and model-47-1.74.h5 seeds the joint model's decoder.

File layout Keras writes (h5py, libver 'earliest'):
    /  attrs: layer_names = [b'conv1', ...]  (fixed-length byte strings; split into layer_names0.. when > 64 KB), backend, keras_version
    /<layer>/  attrs: weight_names = [b'conv1/kernel:0', b'conv1/bias:0']
    /<layer>/<weight path>   one dataset per weight (float32, contiguous)
    (a whole-model file holds the same tree under /model_weights)

What is implemented of the HDF5 File Format Specification (version 0/1 superblock files, i.e. what h5py writes by default):
superblock v0/v1, version-1 object headers with continuation blocks, old-style groups (symbol-table message, v1 B-tree,
local heap, symbol-table nodes), dataspace v1/v2, datatypes fixed-point / IEEE float / fixed-length string (variable-length
strings are returned as None), data layout v1-v3 contiguous / compact / chunked WITHOUT filters, attribute messages v1-v3.
Compressed (filtered) datasets and new-style (link-message / fractal-heap) groups raise NotImplementedError.

The writer emits the same subset (libhdf5's default node sizes: symbol-table nodes of <= 8 links under a v1 B-tree) so that save_weights('x.h5') produces a
file Keras / h5py / h5dump read; tests check it with libhdf5's own tools where present.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(IOError):
    pass


# ----------------------------------------------------------------------------------------------------------------
# reader
# ----------------------------------------------------------------------------------------------------------------

class _Datatype(object):
    def __init__(self, buf, off=0):
        b0 = buf[off]
        self.version, self.cls = b0 >> 4, b0 & 15
        self.bits = buf[off + 1] | (buf[off + 2] << 8) | (buf[off + 3] << 16)
        self.size = struct.unpack_from("<I", buf, off + 4)[0]
        self.dtype = None
        order = ">" if self.bits & 1 else "<"
        if self.cls == 0:                                     # fixed point
            self.dtype = np.dtype("%s%s%d" % (order, "i" if self.bits & 8 else "u", self.size))
        elif self.cls == 1:                                   # IEEE float
            if self.size not in (2, 4, 8):
                raise NotImplementedError("float datatype of %d bytes" % self.size)
            self.dtype = np.dtype("%sf%d" % (order, self.size))
        elif self.cls == 3:                                   # fixed-length string
            self.dtype = np.dtype("S%d" % self.size)
        elif self.cls == 9:                                   # variable length (strings in the global heap): not decoded
            self.dtype = None
        else:
            raise NotImplementedError("HDF5 datatype class %d" % self.cls)


def _dataspace(buf, off=0):
    ver, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    p = off + (8 if ver == 1 else 4)
    if ver not in (1, 2):
        raise NotImplementedError("dataspace message version %d" % ver)
    if ver == 2 and buf[off + 3] == 2:
        return None                                           # null dataspace
    return tuple(struct.unpack_from("<%dQ" % rank, buf, p)) if rank else ()


class H5Object(object):
    """A group or a dataset: .attrs (dict), .keys() / [name] for groups, .read() for datasets."""

    def __init__(self, f, addr):
        self._f, self.addr = f, addr
        self.attrs = {}
        self._btree = self._heap = None
        self._shape = self._dt = self._layout = None
        self._filtered = False
        self._links = None
        self._parse_header()

    # -- object header (version 1) --------------------------------------------------------------------------------
    def _messages(self):
        b = self._f.buf
        a = self.addr
        if b[a:a + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (a file written with libver='latest')")
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", b, a)
        if ver != 1:
            raise Hdf5Error("object header version %d at %d" % (ver, a))
        blocks = [(a + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, p)
                body = p + 8
                if mtype == 0x0010:                           # continuation
                    off, length = struct.unpack_from("<QQ", b, body)
                    blocks.append((off + self._f.base, length))
                else:
                    yield mtype, body, msize, flags
                seen += 1
                p = body + msize

    def _parse_header(self):
        b = self._f.buf
        for mtype, p, size, flags in self._messages():
            if mtype == 0x0011:
                self._btree, self._heap = struct.unpack_from("<QQ", b, p)
            elif mtype == 0x0001:
                self._shape = _dataspace(b, p)
            elif mtype == 0x0003:
                self._dt = _Datatype(b, p)
            elif mtype == 0x0008:
                self._layout = self._parse_layout(p)
            elif mtype == 0x000B:
                self._filtered = True
            elif mtype == 0x000C:
                name, value = self._parse_attribute(p)
                self.attrs[name] = value
            elif mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style groups (link messages): re-save the file with h5py's default libver")

    def _parse_layout(self, p):
        b = self._f.buf
        ver = b[p]
        if ver == 3:
            cls = b[p + 1]
            if cls == 0:
                n = struct.unpack_from("<H", b, p + 2)[0]
                return ("compact", p + 4, n)
            if cls == 1:
                addr, n = struct.unpack_from("<QQ", b, p + 2)
                return ("contiguous", addr, n)
            if cls == 2:
                rank = b[p + 2]
                addr = struct.unpack_from("<Q", b, p + 3)[0]
                dims = struct.unpack_from("<%dI" % rank, b, p + 11)
                return ("chunked", addr, dims)
            raise NotImplementedError("data layout class %d" % cls)
        if ver in (1, 2):
            rank, cls = b[p + 1], b[p + 2]
            q = p + 8
            addr = None
            if cls != 0:
                addr = struct.unpack_from("<Q", b, q)[0]
                q += 8
            dims = struct.unpack_from("<%dI" % rank, b, q)
            q += 4 * rank
            if cls == 0:
                n = struct.unpack_from("<I", b, q)[0]
                return ("compact", q + 4, n)
            if cls == 1:
                return ("contiguous", addr, None)
            return ("chunked", addr, dims)
        raise NotImplementedError("data layout message version %d" % ver)

    def _parse_attribute(self, p):
        b = self._f.buf
        ver = b[p]
        nsize, tsize, ssize = struct.unpack_from("<HHH", b, p + 2)
        q = p + 8
        if ver == 3:
            q += 1                                           # name character set
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = bytes(b[q:q + nsize]).split(b"\0")[0].decode("utf-8")
        q += pad(nsize)
        dt = _Datatype(b, q)
        q += pad(tsize)
        shape = _dataspace(b, q)
        q += pad(ssize)
        if dt.dtype is None or shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        arr = np.frombuffer(b, dt.dtype, n, q).reshape(shape)
        return name, (arr.copy() if shape else arr.reshape(()).copy()[()])

    # -- groups ---------------------------------------------------------------------------------------------------
    @property
    def is_group(self):
        return self._btree is not None

    def _walk_links(self):
        if self._links is not None:
            return self._links
        f, b = self._f, self._f.buf
        links = {}
        if self._btree is None:
            self._links = links
            return links
        h = self._heap + f.base
        if b[h:h + 4] != b"HEAP":
            raise Hdf5Error("bad local heap signature at %d" % h)
        heap_data = struct.unpack_from("<Q", b, h + 24)[0] + f.base

        raw = self._f.raw

        def name_at(off):
            return raw[heap_data + off:raw.index(b"\0", heap_data + off)].decode("utf-8")

        def node(addr):
            a = addr + f.base
            if b[a:a + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, a + 6)[0]
                for i in range(n):
                    noff, oaddr = struct.unpack_from("<QQ", b, a + 8 + 40 * i)
                    links[name_at(noff)] = oaddr + f.base
                return
            if b[a:a + 4] != b"TREE":
                raise Hdf5Error("bad B-tree signature at %d" % a)
            used = struct.unpack_from("<H", b, a + 6)[0]
            for i in range(used):
                node(struct.unpack_from("<Q", b, a + 24 + 8 + 16 * i)[0])
        node(self._btree)
        self._links = links
        return links

    def keys(self):
        return sorted(self._walk_links())

    def __contains__(self, name):
        return name.split("/")[0] in self._walk_links()

    def __getitem__(self, path):
        obj = self
        for part in [p for p in path.split("/") if p]:
            links = obj._walk_links()
            if part not in links:
                raise KeyError(path)
            obj = self._f._object(links[part])
        return obj

    # -- datasets -------------------------------------------------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return None if self._dt is None else self._dt.dtype

    def read(self):
        if self._layout is None or self._dt is None or self._dt.dtype is None:
            raise Hdf5Error("not a readable dataset")
        if self._filtered:
            raise NotImplementedError("compressed / filtered datasets (Keras writes its weights unfiltered)")
        f, b = self._f, self._f.buf
        shape = self._shape or ()
        n = int(np.prod(shape)) if shape else 1
        kind = self._layout[0]
        if kind == "compact":
            return np.frombuffer(b, self._dt.dtype, n, self._layout[1]).reshape(shape).copy()
        if kind == "contiguous":
            if self._layout[1] == UNDEF:
                return np.zeros(shape, self._dt.dtype)          # never written: fill value
            return np.frombuffer(b, self._dt.dtype, n, self._layout[1] + f.base).reshape(shape).copy()
        out = np.zeros(shape, self._dt.dtype)                    # chunked, no filters
        cdims = self._layout[2][:-1]
        rank = len(cdims)

        def node(addr):
            a = addr + f.base
            if b[a:a + 4] != b"TREE":
                raise Hdf5Error("bad chunk B-tree signature at %d" % a)
            level, used = b[a + 5], struct.unpack_from("<H", b, a + 6)[0]
            p = a + 24
            ksize = 8 + 8 * (rank + 1)
            for _ in range(used):
                nbytes, mask = struct.unpack_from("<II", b, p)
                offs = struct.unpack_from("<%dQ" % rank, b, p + 8)
                child = struct.unpack_from("<Q", b, p + ksize)[0]
                p += ksize + 8
                if level > 0:
                    node(child)
                    continue
                chunk = np.frombuffer(b, self._dt.dtype, int(np.prod(cdims)), child + f.base).reshape(cdims)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        if self._layout[1] != UNDEF:
            node(self._layout[1])
        return out


class H5File(H5Object):
    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            self.raw = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, "rb") as fh:
                self.raw = fh.read()
        self.buf = memoryview(self.raw)
        b = self.buf
        start = 0
        while bytes(b[start:start + 8]) != SIGNATURE:
            start = 512 if start == 0 else start * 2
            if start + 8 > len(b):
                raise Hdf5Error("not an HDF5 file")
        ver = b[start + 8]
        if ver not in (0, 1):
            raise NotImplementedError("superblock version %d (written with libver='latest'); Keras / h5py default files are version 0" % ver)
        if b[start + 13] != 8 or b[start + 14] != 8:
            raise NotImplementedError("offsets / lengths of %d / %d bytes" % (b[start + 13], b[start + 14]))
        p = start + 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", b, p)[0]
        root = p + 32
        root_ohdr = struct.unpack_from("<Q", b, root + 8)[0]
        self._cache = {}
        H5Object.__init__(self, self, root_ohdr + self.base)

    def _object(self, addr):
        if addr not in self._cache:
            self._cache[addr] = H5Object(self, addr)
        return self._cache[addr]


def _attr_list(group, name):
    """Keras' load_attributes_from_hdf5_group: `name`, or the chunks name0, name1, ... a long list was split into."""
    if name in group.attrs and group.attrs[name] is not None:
        vals = list(np.atleast_1d(group.attrs[name]))
    else:
        vals, i = [], 0
        while "%s%d" % (name, i) in group.attrs:
            vals.extend(np.atleast_1d(group.attrs["%s%d" % (name, i)]))
            i += 1
    return [v.decode("utf-8") if isinstance(v, bytes) else str(v) for v in vals]


def load_keras_weights(path):
    """{'<layer>/<weight>': ndarray} of a Keras weights file (or the model_weights group of a whole-model file), keyed like the
    repo's .npz checkpoints by the weight's OWN name scope: Keras names a weight '<layer>/<weight>:0', and a nested model or
    wrapper (TimeDistributed, the 'imgcap_caption_td' sub-model of text_generation_model.py:159-189) stores its inner layers'
    weights under the outer layer's group with their inner names ('imgcap_lstm1/kernel:0') -- the last two path components
    are the key, which is what loading "by name" needs."""
    f = H5File(path)
    root = f
    if "layer_names" not in f.attrs and "layer_names0" not in f.attrs and "model_weights" in f:
        root = f["model_weights"]
    out = {}
    for layer in _attr_list(root, "layer_names"):
        g = root[layer]
        for wname in _attr_list(g, "weight_names"):
            parts = wname.split(":")[0].split("/")
            out["/".join(parts[-2:]) if len(parts) >= 2 else "%s/%s" % (layer, parts[0])] = g[wname].read()
    return out


# ----------------------------------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------------------------------

def _pad8(b):
    return b + b"\0" * ((-len(b)) % 8)


def _dt_message(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "f":
        size = dtype.itemsize
        exp, mant, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[size]
        bits = size * 8
        return struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, size) + struct.pack("<HHBBBBI", 0, bits, mant, exp, 0, mant, bias)
    if dtype.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dtype.kind == "i" else 0, 0, 0, dtype.itemsize) + struct.pack("<HH", 0, dtype.itemsize * 8)
    if dtype.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dtype.itemsize)         # null-padded ASCII, like numpy 'S' through h5py
    raise NotImplementedError("dtype %s" % dtype)


def _ds_message(shape):
    rank = len(shape)
    return struct.pack("<BBBBI", 1, rank, 1 if rank else 0, 0, 0) + struct.pack("<%dQ" % rank, *shape) * (2 if rank else 1)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHBBH", mtype, len(body), flags, 0, 0) + body


def _attr_message(name, value):
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind == "f" and arr.dtype.itemsize == 8 and arr.ndim == 0:
        pass
    nm = name.encode("utf-8") + b"\0"
    dt, ds = _dt_message(arr.dtype), _ds_message(arr.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + np.ascontiguousarray(arr).tobytes()
    return _msg(0x000C, body)


class _Writer(object):
    LEAF_K, INT_K = 4, 16                          # libhdf5's defaults: <= 8 links per symbol-table node, <= 32 children per B-tree node

    def __init__(self):
        self.buf = bytearray(96)                   # superblock + root symbol-table entry, filled in at the end

    def alloc(self, data, align=8):
        self.buf.extend(b"\0" * ((-len(self.buf)) % align))
        addr = len(self.buf)
        self.buf.extend(data)
        return addr

    def object_header(self, messages):
        body = b"".join(messages)
        return self.alloc(struct.pack("<BBHII", 1, 0, len(messages), 1, len(body)) + b"\0" * 4 + body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr)
        data = self.alloc(arr.tobytes() or b"\0")
        fill = struct.pack("<BBBB", 2, 2, 2, 0)                              # version 2, allocate late, write if set, undefined
        layout = struct.pack("<BBQQ", 3, 1, data, arr.nbytes)
        return self.object_header([_msg(0x0001, _ds_message(arr.shape)), _msg(0x0003, _dt_message(arr.dtype), 1), _msg(0x0005, fill),
                                   _msg(0x0008, layout)])

    def _tree_node(self, level, children, keys):
        """children: addresses; keys: len(children)+1 heap offsets (key[i] < names of child i <= key[i+1])."""
        node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, level, len(children), UNDEF, UNDEF))
        for i, c in enumerate(children):
            node.extend(struct.pack("<QQ", keys[i], c))
        node.extend(struct.pack("<Q", keys[len(children)]))
        node.extend(b"\0" * (24 + (2 * self.INT_K + 1) * 8 + 2 * self.INT_K * 8 - len(node)))
        return self.alloc(bytes(node))

    def group(self, links, attrs):
        """links: {name: object header address}; returns (object header address, btree address, heap address)."""
        names = sorted(links, key=lambda s: s.encode("utf-8"))
        heap = bytearray(b"\0" * 8)                                           # offset 0: the empty string
        offs = []
        for n in names:
            offs.append(len(heap))
            heap.extend(_pad8(n.encode("utf-8") + b"\0"))
        free_off = len(heap)
        heap.extend(struct.pack("<QQ", 1, 16))                                # one free block closing the heap (next = 1: last)
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap), free_off, heap_data))
        # leaves: symbol-table nodes of <= 2*LEAF_K links, in name order
        per = 2 * self.LEAF_K
        level = [(None, 0)]                                                   # (address, largest-name heap offset) per node of the current level
        leaves = []
        for i in range(0, len(names), per):
            chunk = list(zip(names[i:i + per], offs[i:i + per]))
            snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)))
            for n, o in chunk:
                snod.extend(struct.pack("<QQII", o, links[n], 0, 0) + b"\0" * 16)
            snod.extend(b"\0" * (8 + per * 40 - len(snod)))
            leaves.append((self.alloc(bytes(snod)), chunk[-1][1]))
        nodes, lvl = leaves, 0
        while True:                                                           # B-tree levels until one node remains
            fan = 2 * self.INT_K
            parents = []
            for i in range(0, max(len(nodes), 1), fan):
                grp = nodes[i:i + fan]
                first_key = 0 if i == 0 else nodes[i - 1][1]
                keys = [first_key] + [k for _, k in grp]
                parents.append((self._tree_node(lvl, [a for a, _ in grp], keys), grp[-1][1] if grp else 0))
            nodes, lvl = parents, lvl + 1
            if len(nodes) == 1:
                break
        tree_addr = nodes[0][0]
        msgs = [_msg(0x0011, struct.pack("<QQ", tree_addr, heap_addr))] + [_attr_message(k, v) for k, v in attrs.items()]
        return self.object_header(msgs), tree_addr, heap_addr

    def finish(self, root, tree, heap):
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INT_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", tree, heap)
        self.buf[0:96] = sb
        return bytes(self.buf)


def _chunked_attrs(name, values, limit=64512):
    """Keras' save_attributes_to_hdf5_group: one attribute, or name0, name1, ... when the array exceeds the object header's
    64 KB message limit."""
    arr = np.array(values) if len(values) else np.zeros((0,), "S1")
    if arr.nbytes <= limit:
        return {name: arr}
    parts = 2
    while max(c.nbytes for c in np.array_split(arr, parts)) > limit:
        parts += 1
    return {"%s%d" % (name, i): c for i, c in enumerate(np.array_split(arr, parts))}


# position of a weight inside its Keras layer's `weights` list (Conv2D / Dense: kernel, bias; LSTM: kernel, recurrent_kernel, bias;
# BatchNormalization: gamma, beta, moving_mean, moving_variance; Embedding: embeddings).  Keras' by-name loader matches the GROUP by
# name and then assigns weight_values[i] to layer.weights[i] BY POSITION, so a file must list them in this order.
_KERAS_WEIGHT_RANK = {"kernel": 0, "recurrent_kernel": 1, "bias": 2, "gamma": 0, "beta": 1, "moving_mean": 2, "moving_variance": 3,
                      "embeddings": 0}


def keras_weight_order(items):
    """[(weight name, array)] of one layer in the order of Keras' layer.weights (stable for names Keras does not know)."""
    return sorted(items, key=lambda it: _KERAS_WEIGHT_RANK.get(it[0], 99))


def save_keras_weights(path, weights, layer_order=None, layer_groups=None, group_member_order=None):
    """Write {'<layer>/<weight>': ndarray} as a Keras weights file: /<layer>/<layer>/<weight>:0 datasets, weight_names and
    layer_names attributes (fixed-length byte strings), float32 data.  layer_order: Keras lists layers in model order; default
    is first-seen order of `weights`.
    layer_groups: {inner layer: outer layer} for layers that live inside a wrapper or nested model -- Keras stores those under the
    OUTER layer's group with the inner names (the joint model's decoder: TimeDistributed(caption_model, name='imgcap_caption_td'),
    dense_img_cap/dense_model.py:1554-1560: /imgcap_caption_td/imgcap_lstm1/kernel:0 with weight_names 'imgcap_lstm1/kernel:0'),
    which is where the reference's load_weights(by_name=True) looks for them.
    group_member_order: {outer layer: [inner layers in the order of the wrapped model's `weights` list]} -- for a Model that is its
    trainable weights in layer order followed by the non-trainable ones (the frozen embedding of the caption decoder comes LAST);
    default: the order the members appear in `layer_order`."""
    w = _Writer()
    layers = {}
    for key, arr in weights.items():
        layer, name = key.split("/", 1)
        layers.setdefault(layer, []).append((name, np.asarray(arr, np.float32)))
    layers = {l: keras_weight_order(items) for l, items in layers.items()}
    order = list(layer_order) if layer_order is not None else list(layers)
    groups = dict(layer_groups or {})
    outer_order, members = [], {}
    for layer in order:                                   # outer groups in first-seen order of their members
        g = groups.get(layer, layer)
        if g not in members:
            members[g] = []
            outer_order.append(g)
        members[g].append(layer)
    for g, want in (group_member_order or {}).items():
        if g in members:
            members[g] = [l for l in want if l in members[g]] + [l for l in members[g] if l not in want]
    top = {}
    for g in outer_order:
        inner_groups, names = {}, []
        for layer in members[g]:
            items = layers.get(layer, [])
            if not items:
                continue
            inner = {("%s:0" % n): w.dataset(a) for n, a in items}
            inner_groups[layer], _, _ = w.group(inner, {})
            names += [("%s/%s:0" % (layer, n)).encode("utf-8") for n, _ in items]
        top[g], _, _ = w.group(inner_groups, _chunked_attrs("weight_names", names))
    attrs = _chunked_attrs("layer_names", [l.encode("utf-8") for l in outer_order])
    attrs.update({"backend": np.array(b"tensorflow"), "keras_version": np.array(b"2.1.6")})
    root, tree, heap = w.group(top, attrs)
    data = w.finish(root, tree, heap)
    with open(path, "wb") as fh:
        fh.write(data)
