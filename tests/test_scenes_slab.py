"""The rank-local scene generators used by bench.py --scaling strong produce exactly the slab of the global scene."""
import numpy as np
import pytest

from polystokes_amd import partition, scenes


@pytest.mark.parametrize("name", ["coil", "spheres", "cavity"])
@pytest.mark.parametrize("world", [2, 3])
def test_scene_slab_equals_cut_of_global_scene(name, world):
    n = 48
    glob, p = getattr(scenes, name)(n)
    for rank in range(world):
        loc, pl, sl = scenes.scene_slab(name, n, world, rank)
        ref = partition.local_scene(glob, sl)
        assert (loc.nx, loc.ny, loc.nz) == (ref.nx, ref.ny, ref.nz)
        for a in ("surface", "collision", "viscosity"):
            assert np.array_equal(getattr(loc, a), getattr(ref, a)), (name, rank, a)
        for a in range(3):
            assert np.array_equal(loc.vel[a], ref.vel[a]) and np.array_equal(loc.collisionvel[a], ref.collisionvel[a]), (name, rank, a)
        assert pl.tileSize == p.tileSize
