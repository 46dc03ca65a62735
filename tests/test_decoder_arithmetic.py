"""The integer identities the decoders' inner loops rest on (csrc/kernels/decompress_blocks.inc: dec_match_fields,
dec_literal_run, dec_wrap1 / dec_wrap2, the per-lane tables of idx mod off; decompress_stream.inc: the two-plane
window, the scan's beat), restated in Python with 32-bit wrap-around and checked exhaustively against the plain
form of the rule (lzs-decompression.c:217-233, 238-342).  No GPU: these are the kernels' host-checkable halves."""
import itertools

M32 = 0xFFFFFFFF


def u32(x):
    return x & M32


def ffbh(x):                     # v_ffbh_u32: leading zeros, 0xFFFFFFFF for 0
    return M32 if x == 0 else 32 - x.bit_length()


def match_fields_plain(top):
    """1 s ooooooo[oooo] cccc at the top of a 32-bit word, field by field."""
    short = (top >> 30) & 1
    o = (top >> 23) & 0x7F if short else (top >> 19) & 0x7FF
    used = 9 if short else 13
    code = ((top >> 19) if short else (top >> 15)) & 0xF
    length = 2 + (code >> 2) if code < 0xC else code - 7
    width = 2 if code < 0xC else 4
    return short, o, used, code, length, width


def match_fields_kernel(top):
    """dec_match_fields: a short token moved down four bits has its fields where a long one has them."""
    sign = M32 if (u32(top << 1) >> 31) else 0          # (int32)(top << 1) >> 31
    short4 = sign & 4
    norm = (top & 0x3FFFFFFF) >> short4
    o = norm >> 19
    code = (norm >> 15) & 0xF
    a, b = ((norm >> 17) & 3) + 2, code - 7              # signed max
    return short4, o, code, max(a, b)


def test_match_fields_every_token_prefix():
    # every value of the 17 bits a match token can occupy below its type bit, two fillers for the rest
    for bits17, filler in itertools.product(range(1 << 17), (0, 0x3FFF)):
        top = 0x80000000 | (bits17 << 14) | filler
        short, o, used, code, length, width = match_fields_plain(top)
        short4, o2, code2, len2 = match_fields_kernel(top)
        assert short4 == 4 * short and 13 - short4 == used
        assert (o2, code2, len2) == (o, code, length), hex(top)
        assert (15 if code2 < 0xC else 17) - short4 == used + width          # needB of the second token


def literal_run_plain(bits64, limit):
    n = 0
    while n < limit and not (bits64 >> (63 - 9 * n)) & 1:
        n += 1
    return n


def literal_run_kernel(top, low, short_form):
    if short_form:
        lead = min(ffbh(top & 0x80402010), 36)
    else:
        lead = min(ffbh(top & 0x80402010), ffbh(low & 0x08040200) | 32, 64)
    return (lead * 57) >> 9


def test_literal_run_counts():
    import random
    rng = random.Random(7)
    cases = [0, M32 << 32 | M32, 1 << 63, 1 << 54, 1 << 45, 1 << 36, 1 << 27, 1 << 18, 1 << 9, 1]
    cases += [rng.getrandbits(64) & ~(1 << 63) for _ in range(20000)]        # (a run starts with a literal: type bit 0)
    cases += [rng.getrandbits(64) & rng.getrandbits(64) & rng.getrandbits(64) for _ in range(20000)]   # sparse ones: long runs
    for v in cases:
        top, low = v >> 32, v & M32
        assert literal_run_kernel(top, low, False) == literal_run_plain(v, 7), hex(v)
        assert literal_run_kernel(top, low, True) == literal_run_plain(v, 4), hex(v)
    for x in range(0, 97):                                                   # the divisions by nine: have / 9, lead / 9
        assert (x * 57) >> 9 == x // 9
    for left9 in range(0, 400):                                              # literals that start inside the segment
        assert (min(left9, 72) * 57) >> 9 == min(left9 // 9, 8)


def test_ring_positions_by_min():
    R = 2304                                                                 # kDecRing
    wrap1 = lambda x: min(x, u32(x - R))
    wrap2 = lambda x: min(x, u32(x - R), u32(x - 2 * R))
    for x in range(0, 2 * R):
        assert wrap1(x) == x % R
    for x in range(0, 3 * R):
        assert wrap2(x) == x % R
    # the source of a copy: cpos + k + (R - off), cpos < R, k <= 15, 1 <= off <= 2047
    for cpos in (0, 1, 15, 2047, 2048, R - 16, R - 1):
        for k in range(16):
            for off in (1, 2, 15, 16, 255, 2046, 2047):
                assert wrap2(cpos + k + (R - off)) == (cpos + k - off) % R


def test_windows_start_as_zeros():
    """A copy from before out[0] must read zero: with a window of 2304 bytes that starts as zeros, the position it
    reads has not been written as long as fewer than 2304 bytes are out -- and it can only lie before out[0] while
    fewer than 2047 are."""
    R = 2304
    for count in range(0, 2047):                     # bytes written so far (positions 0 .. count + 15 may be this step's)
        for off in (count + 1, count + 2, 2047):     # sources before out[0]
            if off > 2047:
                continue
            for k in range(min(off, 16)):
                src = count + k - off                # negative: before out[0]
                if src >= 0:
                    continue
                ring = src % R
                assert ring >= count + 16 or ring >= R - 2047        # never a position this stream has written
                assert ring > count + 15


def test_idx_mod_off_tables():
    for j in range(8):
        mod_lo = 0
        for e in range(9):
            mod_lo |= (j if e in (0, 8) else j % e) << (3 * e)
        mod_hi = 0
        for e in range(16):
            mod_hi |= ((j + 8) if e == 0 else (j + 8) % e) << (4 * e)
        for off in range(0, 2048):
            k0 = (mod_lo >> (3 * min(off, 8))) & 7
            k1 = (mod_hi >> (4 * (off & 15))) & 15 if off < 16 else j + 8
            assert k0 == (j if off == 0 else j % off)
            assert k1 == (j + 8 if off == 0 else (j + 8) % off)


def test_two_plane_window_entries():
    """Entry i of the segment decoder's window: eight bits in lo[i], four in nibble i / 1024 of hi[i % 1024]; sixteen
    consecutive entries never share a byte of `hi`, and a store keeps the other nibble."""
    import random
    rng = random.Random(3)
    lo, hi = [0] * 2048, [0] * 1024
    model = [0] * 2048

    def store(i, val):
        nib = (i >> 10) << 2
        lo[i] = val & 0xFF
        hi[i & 1023] = (hi[i & 1023] & ~(15 << nib) & 0xFF) | ((val >> 8) << nib)

    def load(i):
        return lo[i] | (((hi[i & 1023] >> ((i >> 10) << 2)) & 15) << 8)

    for _ in range(20000):
        start, n = rng.randrange(2048), rng.randrange(1, 17)
        idx = [(start + t) & 2047 for t in range(n)]
        assert len({i & 1023 for i in idx}) == n                             # one lane per byte of `hi` in a step
        for i in idx:
            val = rng.randrange(256) if rng.random() < 0.3 else 0x800 | rng.randrange(1, 2048)
            store(i, val)
            model[i] = val
        probe = rng.randrange(2048)
        assert load(probe) == model[probe]
    assert all(load(i) == model[i] for i in range(2048))


def test_scan_beat_never_runs_dry():
    """The scan's input window (32 words per lane, eight asked for on every eighth trip, stored eight trips later if the
    eight words they replace have been used): a lane that takes at most one word a trip always finds its word stored,
    and no word is overwritten before it is used -- for any pattern of taking."""
    import random
    rng = random.Random(11)
    W = 32
    for trial in range(300):
        p_take = rng.choice((0.0, 0.05, 0.3, 0.7, 1.0))
        wi, lc, pending, pl = 1, W, False, 0
        stored_to = W                                  # words [.., stored_to) have been stored at some time
        slot_holds = list(range(W))                    # which word each slot holds
        for tick in range(4000):
            if tick % 8 == 0:
                if pending and wi + W >= pl + 8:
                    for w in range(pl, pl + 8):
                        assert slot_holds[w % W] < wi, "a word not yet used is overwritten"
                        slot_holds[w % W] = w
                    stored_to = pl + 8
                    pending = False
                if not pending:
                    pl, lc, pending = lc, lc + 8, True
            if rng.random() < p_take or (p_take > 0 and tick % 97 < 40):      # bursts of a word a trip
                wi += 1
            assert wi < stored_to and slot_holds[wi % W] == wi, "the word to feed next is not in the window"
