"""The importer's building blocks (scope row f1): OBJ / MTL reader, image decoders, importIntoScene."""
import json
import os

import numpy as np
import pytest

from wurblpt_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ_DIR = os.path.join(ROOT, "tests", "golden", "obj")


def test_obj_reader_matches_the_vendored_tinyobjloader(golden, tmp_path):
    """include/wurblpt/objreader.hpp against what the reference's own parser (tiny_obj_loader.h, compiled
    by oracle/ref_probe.cpp with the importer's configuration) produces for tests/golden/obj/cases.obj:
    every float bit (its number parser is restated), index forms incl. relative indices, triangulation
    of quads (shorter diagonal) and of a pentagon, a concave hexagon and a heptagon in the xz plane
    (its ear clipping), shape boundaries at `g` / `o`, material switches inside a shape, an unknown
    material, MTL values incl. d / Tr precedence and texture options."""
    out = str(tmp_path / "obj.json")
    assert host.lib().wpt_host_obj_dump(os.path.join(OBJ_DIR, "cases.obj").encode(), out.encode()) == 0
    mine = json.load(open(out))
    assert len(mine) == 10
    for key, value in mine.items():
        assert value == golden.raw[key], key
    assert golden.raw["obj_shape_index_counts"] == [21, 12, 9, 15, 18, 3]
