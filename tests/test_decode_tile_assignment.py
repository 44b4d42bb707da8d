"""The tile arithmetic of decode_body (vattention_amd/csrc/decode_body.h), restated on the CPU and checked exhaustively on small sizes: every
32-key tile of a sequence is processed by exactly one wave of exactly one piece, the sequence's last tile (the only ragged one, the one
that holds the appended row) is the LAST tile of the wave that owns it and is taken out of that wave's steady-state loop — for contiguous
pieces (every decomposition but one) and for striped pieces (the single-sequence split launch: piece s = tiles s, s + S, ...)."""
import pytest

W = 4      # DC_WAVES


def wave_tiles(ntiles, tile_begin, tile_end, tstride, wave, special_last):
    """(steady-state tiles, ragged last tile or None) of one wave — the statements of decode_body after `const int last_tile`."""
    last_tile = ntiles - 1
    wstep = W * tstride
    first = tile_begin + wave * tstride
    own_last = special_last and last_tile >= first and last_tile < tile_end and (last_tile - first) % wstep == 0
    loop_end = last_tile if own_last else tile_end
    steady = list(range(first, loop_end, wstep)) if first < loop_end else []
    return steady, (last_tile if own_last else None)


def piece_range(ntiles, S, s, striped):
    """[tile_begin, tile_end) and the stride of piece s of S (split mode of decode_body)."""
    if striped:
        return s, ntiles, S
    per = (ntiles + S - 1) // S
    tb = s * per
    return tb, min(ntiles, tb + per), 1


@pytest.mark.parametrize("striped", [False, True], ids=["contiguous", "striped"])
@pytest.mark.parametrize("special_last", [False, True], ids=["full_last_tile", "ragged_or_appended_last_tile"])
def test_every_tile_once_and_the_last_tile_last(striped, special_last):
    for ntiles in range(1, 90):
        for S in list(range(1, 34)) + [48, 64, 96, 128]:
            seen = [0] * ntiles
            ragged_owner = 0
            for s in range(S):
                tb, te, st = piece_range(ntiles, S, s, striped)
                for wave in range(W):
                    steady, last = wave_tiles(ntiles, tb, te, st, wave, special_last)
                    for t in steady:
                        assert 0 <= t < ntiles and not (special_last and t == ntiles - 1), (ntiles, S, s, wave, t)
                        seen[t] += 1
                    if steady and len(steady) > 1:
                        assert all(b - a == W * st for a, b in zip(steady, steady[1:]))
                    if last is not None:
                        assert not steady or steady[-1] < last       # the ragged tile comes after everything else of its wave
                        seen[last] += 1
                        ragged_owner += 1
            assert seen == [1] * ntiles, (ntiles, S, seen)
            assert ragged_owner == (1 if special_last else 0)


def test_striped_pieces_are_balanced_and_interleaved():
    for ntiles in (1, 5, 64, 4096, 4097):
        for S in (1, 3, 64, 96):
            sizes = []
            for s in range(S):
                tb, te, st = piece_range(ntiles, S, s, True)
                n = sum(len(wave_tiles(ntiles, tb, te, st, w, False)[0]) for w in range(W))
                sizes.append(n)
            assert sum(sizes) == ntiles and max(sizes) - min(sizes) <= 1
            assert sizes == sorted(sizes, reverse=True)        # the non-empty pieces are the first min(S, ntiles): their records are contiguous
