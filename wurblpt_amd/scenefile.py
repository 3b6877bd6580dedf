"""A flattened scene (wpt_scene_desc + camera, include/wurblpt_hip.h) as one file that can be mapped back.

Why: with N processes on one node (one per GPU) every rank needs the same read-only scene description, and building
it is the expensive part of start-up (the reference's BVH build over 10 M hitables takes most of a minute on all
cores, bvh.hpp:93-270).  Rank 0 builds and `save`s it (to /dev/shm, which is memory), the other ranks `load` it:
numpy maps the file, the pointers of the description point into the mapping, and wpt_scene_upload copies from there
to the GPU -- no rank but the first runs the builder, nothing is unpickled or recomputed.

Layout: 16-byte magic and header length, a JSON header (counts, the two structs as hex, one [offset, bytes] pair per
array), then the arrays, each starting at a multiple of 64 bytes.  It is a transport between processes of one run of
one build of this package, not an interchange format: `load` refuses a file written for another ABI version."""
import ctypes as C
import json
import os

import numpy as np

from . import _abi

MAGIC = b"WPTSCENE"

# pointer member -> (count member or callable, element struct); the bytes of an array are count * sizeof(element)
_ARRAYS = [
    ("nodes", "node_count", _abi.BvhNode), ("tri_geom", "tri_count", _abi.TriGeom), ("tri_attr", "tri_count", _abi.TriAttr),
    ("instances", "instance_count", _abi.Instance), ("materials", "material_count", _abi.Material),
    ("textures", "texture_count", _abi.Texture), ("texels", "texel_bytes", C.c_uint8), ("hotspots", "hotspot_count", _abi.Hotspot),
    ("spheres", "sphere_count", _abi.Sphere), ("rgl_brdfs", "rgl_count", _abi.RglBrdf), ("rgl_data", "rgl_data_count", C.c_float),
    ("animations", "animation_count", _abi.Animation), ("keyframes", "keyframe_count", _abi.Keyframe),
]
_ENV_ARRAYS = ["M", "Ms", "Mcs"]  # N * N four-byte values each, or NULL


def _address(pointer):
    if pointer is None:
        return 0
    if isinstance(pointer, int):
        return pointer
    return C.cast(pointer, C.c_void_p).value or 0


def save(scene, path):
    """Writes the description and camera of a host scene (wurblpt_amd.host.HostScene or a loaded one) to `path`."""
    d = scene.desc.contents
    blobs, table, offset = [], {}, 0

    def add(name, address, nbytes):
        nonlocal offset
        if not address or nbytes == 0:
            table[name] = [0, 0]
            return
        offset = (offset + 63) // 64 * 64
        table[name] = [offset, nbytes]
        blobs.append((offset, (C.c_uint8 * nbytes).from_address(address)))
        offset += nbytes

    for member, count, element in _ARRAYS:
        add(member, _address(getattr(d, member)), int(getattr(d, count)) * C.sizeof(element))
    bins = int(d.envmap.N) * int(d.envmap.N) * 4
    for member in _ENV_ARRAYS:
        add("envmap." + member, _address(getattr(d.envmap, member)), bins)
    camera = C.cast(scene.camera, C.POINTER(_abi.Camera)).contents
    header = json.dumps({
        "abi_version": int(d.abi_version), "width": scene.width, "height": scene.height, "name": scene.name,
        "bvh_levels": int(getattr(scene, "bvh_levels", 0)),
        "desc": bytes(d).hex(), "camera": bytes(camera).hex(), "arrays": table, "data_bytes": offset,
    }).encode()
    data_start = (16 + len(header) + 63) // 64 * 64
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(MAGIC + len(header).to_bytes(8, "little"))
        f.write(header)
        for at, blob in blobs:
            f.seek(data_start + at)
            f.write(blob)
        f.truncate(data_start + offset)
    os.replace(tmp, path)  # readers see the whole file or none of it


class FileScene:
    """A scene description whose arrays live in a mapped file; what wurblpt_amd.device.DeviceScene, bench.py and the
    tests' oracle loader need of a HostScene (desc, camera, d, width, height, name, set_envmap_tables)."""

    def __init__(self, path):
        with open(path, "rb") as f:
            head = f.read(16)
            if head[:8] != MAGIC:
                raise RuntimeError("%s is not a scene file of this package" % path)
            header = json.loads(f.read(int.from_bytes(head[8:], "little")).decode())
        data_start = (16 + int.from_bytes(head[8:], "little") + 63) // 64 * 64
        self._desc = _abi.SceneDesc.from_buffer_copy(bytes.fromhex(header["desc"]))
        if int(self._desc.abi_version) != int(header["abi_version"]):
            raise RuntimeError("%s: damaged header" % path)
        self._camera = _abi.Camera.from_buffer_copy(bytes.fromhex(header["camera"]))
        size = os.path.getsize(path)
        if size < data_start + int(header["data_bytes"]):
            raise RuntimeError("%s is shorter than its header says" % path)
        # copy-on-write mapping: ctypes wants writable memory for from_buffer; nothing writes to it
        self._map = np.memmap(path, dtype=np.uint8, mode="c", offset=data_start, shape=(max(1, int(header["data_bytes"])),))
        base = self._map.ctypes.data

        def pointer(name):
            at, nbytes = header["arrays"][name]
            return base + at if nbytes else None

        for member, _, element in _ARRAYS:
            address = pointer(member)
            if member == "texels":
                self._desc.texels = address
            else:
                setattr(self._desc, member, C.cast(C.c_void_p(address), C.POINTER(element)))
        for member in _ENV_ARRAYS:
            setattr(self._desc.envmap, member, pointer("envmap." + member))
        self.desc = C.pointer(self._desc)
        self.camera = C.pointer(self._camera)
        self.width, self.height, self.name = header["width"], header["height"], header["name"]
        self.bvh_levels = header["bvh_levels"]
        self.path = path

    @property
    def d(self):
        return self._desc

    def set_envmap_tables(self, M, Ms, Mcs):
        self._env_tables = (M, Ms, Mcs)
        e = self._desc.envmap
        e.M, e.Ms, e.Mcs = M.ctypes.data, Ms.ctypes.data, Mcs.ctypes.data


def load(path):
    return FileScene(path)


def build_once(build, path, rank, barrier, broadcast=None):
    """N ranks, one builder: rank 0 runs build() and saves the scene to `path`, every other rank waits and maps the
    file.  Returns the scene of this rank.  `barrier` is a callable (torch.distributed.barrier); with `broadcast`
    (torch.distributed.broadcast_object_list) rank 0 tells the others whether it succeeded, so that a failed build or a
    full /dev/shm raises on every rank instead of leaving them at a barrier until the process group times out.  The
    file is removed as soon as every rank has mapped it (a mapping outlives its name), so nothing stays behind in
    /dev/shm when a run dies later."""
    error = None
    scene = None
    if rank == 0:
        try:
            scene = build()
            save(scene, path)
        except BaseException as e:  # the other ranks must hear of it whatever it was
            error = "%s: %s" % (type(e).__name__, e)
            try:
                os.remove(path)
            except OSError:
                pass
    if broadcast is not None:
        box = [error]
        broadcast(box, src=0)
        error = box[0]
    else:
        barrier()
    if error is not None:
        raise RuntimeError("rank 0 could not build or save the scene: %s" % error)
    if rank != 0:
        scene = load(path)
    barrier()          # every rank holds its mapping
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    barrier()          # the name is gone when any rank returns
    return scene
