"""Host-side description of the tiled SAD-surface layout written by xmi_sad_surface* (pure indexing)."""
import numpy as np

NBLK = (64, 16, 4, 1)


def row_elems(rng, level):
    """elements per (ctu, dy) row of the main part (dx < +R) of a level's plane"""
    return (rng // 8) * NBLK[level] * 16


def row_pitch(rng, level):
    """elements between the starts of two rows: the payload plus unwritten padding (xmi::surface_row_pitch)"""
    esize = 2 if level < 2 else 4
    nbytes = row_elems(rng, level) * esize
    return (nbytes + max(128, nbytes // 32)) // esize


def plane_elems(nctu, rng, level):
    return nctu * (2 * rng + 1) * (row_pitch(rng, level) + NBLK[level])


def _z16(by16, bx16):
    return (((by16 >> 1) * 2 + (bx16 >> 1)) << 2) | ((by16 & 1) * 2 + (bx16 & 1))


def slot_of_block(level):
    """slot index inside a tile for each raster-order block of the level"""
    if level == 0:
        return np.array([((by & 1) * 2 + (bx & 1)) * 16 + _z16(by >> 1, bx >> 1) for by in range(8) for bx in range(8)])
    if level == 1:
        return np.array([_z16(by, bx) for by in range(4) for bx in range(4)])
    return np.arange(NBLK[level])


def untile(plane, nctu, rng, level):
    """tiled plane (flat array) -> natural [ctu][blk (raster)][dy+R][dx+R] array"""
    nmv, t, nb = 2 * rng + 1, rng // 8, NBLK[level]
    plane = np.asarray(plane).reshape(-1)
    pitch = row_pitch(rng, level)
    nmain = nctu * nmv * pitch
    full = plane[:nmain].reshape(nctu, nmv, pitch)[:, :, :t * nb * 16].reshape(nctu, nmv, t, nb, 16).transpose(0, 3, 1, 2, 4).reshape(nctu, nb, nmv, 2 * rng)
    last = plane[nmain:nmain + nctu * nmv * nb].reshape(nctu, nmv, nb).transpose(0, 2, 1)[..., None]
    nat = np.concatenate([full, last], axis=3)
    return np.ascontiguousarray(nat[:, slot_of_block(level)])
