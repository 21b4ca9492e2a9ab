"""Ocroot::write / read (src/Octree.cpp:178-201): byte layout of the reference's raw chunk dump, round trip, validation."""
import struct

import numpy as np
import pytest


def test_byte_layout_matches_the_reference_struct_dump(svo, tmp_path):
    """Hand-derived from Ocroot (src/Octree.h:56-76) on x86-64: position @0, size @12, depth @16, trees @24, twigs @32,
    treestoragesize @40, twigstoragesize @48, modified @56 (64-byte header = sizeof(Ocroot) - 2 pointers), then nodes, then bricks."""
    tree = np.array([(2 << 30) | 1, (3 << 30) | 0, 1 << 30 | 4, 0, 0, 0, 0, 0, 0], np.uint32)
    twig = np.arange(64, dtype=np.uint16)
    path = str(tmp_path / "chunk.bin")
    svo.chunk_write(path, dict(position=(128.0, 0.0, -256.0), size=128.0, depth=3, tree=tree, twig=twig), 32, 16)
    raw = open(path, "rb").read()
    assert len(raw) == 64 + 9 * 4 + 128
    assert struct.unpack_from("<3f", raw, 0) == (128.0, 0.0, -256.0)
    assert struct.unpack_from("<f", raw, 12)[0] == 128.0 and struct.unpack_from("<I", raw, 16)[0] == 3
    assert struct.unpack_from("<4Q", raw, 24) == (9, 1, 32, 16)
    assert raw[56:64] == bytes(8)
    assert np.array_equal(np.frombuffer(raw, np.uint32, 9, 64), tree)
    assert np.array_equal(np.frombuffer(raw, np.uint16, 64, 64 + 36), twig)


def test_round_trip_of_a_generated_chunk(svo, oracle, tmp_path):
    W = svo.World.generate(1, 1, 1, 128, 6)
    c = W.chunk(0)
    path = str(tmp_path / "c.bin")
    svo.chunk_write(path, c)
    back = svo.chunk_read(path)
    assert back["position"] == c["position"] and back["size"] == c["size"] and back["depth"] == c["depth"]
    assert np.array_equal(back["tree"], c["tree"]) and np.array_equal(back["twig"], c["twig"])
    W2 = svo.World.create([back], 1, 1, 1, 128)
    assert W2.info.total_trees == W.info.total_trees
    # a read-back world marches like the original (CPU: through the oracle over the same arrays)
    rng = np.random.default_rng(1)
    o = (rng.random((500, 3)) * 128).astype(np.float32); d = rng.normal(size=(500, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a = oracle.OracleWorld.from_chunks([c], 1, 1, 1, 128).trace_rays(o, d)
    b = oracle.OracleWorld.from_chunks([back], 1, 1, 1, 128).trace_rays(o, d)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_truncated_or_inconsistent_files_are_rejected(svo, tmp_path):
    tree = np.array([1 << 30 | 2], np.uint32)
    path = str(tmp_path / "c.bin")
    svo.chunk_write(path, dict(position=(0, 0, 0), size=128.0, depth=2, tree=tree, twig=np.zeros(0, np.uint16)))
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-2])
    with pytest.raises(svo.SvoError) as e:
        svo.chunk_read(path)
    assert e.value.code == -4
    open(path, "wb").write(raw[:40])
    with pytest.raises(svo.SvoError):
        svo.chunk_read(path)
    with pytest.raises(svo.SvoError):
        svo.chunk_read(str(tmp_path / "missing.bin"))
