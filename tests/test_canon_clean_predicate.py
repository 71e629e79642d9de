"""CPU (-m "not gpu"): the byte-parallel "clean window" test of the relaxed body canonicaliser (csrc/canon.hip.h,
clean_prefix) restated on 32-bit words.  It may call a clean window dirty (the window then takes the exact path) but
never a dirty one clean: for every byte pair the exact rule flags — TAB anywhere, SP in front of SP / TAB / CR — the
word test must flag it too, whatever the neighbouring bytes are (no carries between bytes)."""
import random

M32 = 0xFFFFFFFF


def word_flags(x: int, up: int) -> int:
    """bit 7 of byte j set <=> byte j of x makes the window not clean; `up` = the dword that follows x."""
    hi = (((x & 0x7F7F7F7F) + 0x5F5F5F5F) | x) & M32                 # bit 7 of byte j: byte j > 0x20
    hi_up = (((up & 0x7F7F7F7F) + 0x5F5F5F5F) | up) & M32
    nh = ((hi >> 8) | (hi_up << 24)) & M32                           # byte j+1 under byte j
    wl = ~(hi | (x << 6) | (x << 5)) & M32                            # <= 0x20 with bits 1 and 2 clear: SP, TAB, seven control codes
    v = wl & ~(nh & (x << 2)) & M32                                   # ... unless SP-like (bit 5) with a byte > 0x20 behind it
    return v & 0x80808080


def exact_dirty(c: int, nxt: int) -> bool:
    return c == 0x09 or (c == 0x20 and nxt in (0x20, 0x09, 0x0D))


def test_every_byte_pair_in_every_position():
    rng = random.Random(5)
    false_positive_bytes = set()
    for c in range(256):
        for nxt in range(256):
            for pos in range(4):
                other = [rng.randrange(256) for _ in range(8)]
                b = other[:4]
                u = other[4:]
                b[pos] = c
                if pos < 3:
                    b[pos + 1] = nxt
                else:
                    u[0] = nxt
                x = b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24
                up = u[0] | u[1] << 8 | u[2] << 16 | u[3] << 24
                flagged = (word_flags(x, up) >> (8 * pos + 7)) & 1
                if exact_dirty(c, nxt):
                    assert flagged, (hex(c), hex(nxt), pos)
                elif flagged:
                    false_positive_bytes.add(c)
    # what the test gives up: SP in front of any byte <= 0x20 (LF, NUL, ...) and seven control codes that look like WSP
    assert false_positive_bytes == {0x00, 0x01, 0x08, 0x10, 0x11, 0x18, 0x19, 0x20}


def test_ordinary_text_is_clean():
    text = b"The quick brown fox, jumps over the lazy dog; 0123456789.\r\nNext line here\r\n"
    text = text * 4
    for i in range(0, len(text) - 8, 4):
        x = int.from_bytes(text[i:i + 4], "little")
        up = int.from_bytes(text[i + 4:i + 8], "little")
        assert word_flags(x, up) == 0, text[i:i + 8]
