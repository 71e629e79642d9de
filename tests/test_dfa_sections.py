"""ZKE_DFA_DECODE_FAIL carries the SECTION of the regex-automata blob at which dense::DFA::from_bytes gives up
(core/src/regex.rs:32-33) as its `detail` — ZKE_D_DFA_LABEL .. ZKE_D_DFA_QUITSET for the forward blob, + 10 for the reverse one.
The sections the two regex-automata-written blobs do not pin (the unanchored start block, accelerators, the quit set:
DESIGN.md §4) have codes of their own, so the first real `dfa::regex::Regex` pair that fails to load says where the recalled
layout is wrong instead of a bare "decode failed".

Every blob here is cut or damaged at a known place of the layout (computed from the blob's own fields); the oracle (CPU tier) and
the engine's registry (GPU tier) must name that section."""
import struct

import pytest

from zkemail_rs_amd import _abi as A
from zkemail_rs_amd import regex_compile as rc

from test_regex_automata_blobs import blob

NONE32 = 0xFFFFFFFF


def section_starts(b: bytes):
    """Byte offset at which each section of the blob begins (the layout of tests/test_regex_automata_blobs.py::parse)."""
    def u32(o):
        return struct.unpack_from("<I", b, o)[0]
    s = {A.D_DFA_LABEL: 0, A.D_DFA_ENDIAN_VERSION: 32, A.D_DFA_FLAGS: 44, A.D_DFA_TRANSITIONS: 48}
    state_len, stride2 = u32(48), u32(52)
    o = 48 + 8 + 256 + 4 * (state_len << stride2)
    s[A.D_DFA_START_TABLE] = o
    stride, npat = u32(o + 4 + 256), u32(o + 4 + 256 + 4)
    o += 4 + 256 + 16 + 4 * (2 * stride + stride * (0 if npat == NONE32 else npat))
    s[A.D_DFA_MATCH_STATES] = o
    ms = u32(o)
    o += 4 + 8 * ms + 4
    ids = u32(o)
    o += 4 + 4 * ids
    s[A.D_DFA_SPECIAL] = o
    o += 32
    s[A.D_DFA_ACCELS] = o
    acc = u32(o)
    o += 4 + 8 * acc
    s[A.D_DFA_QUITSET] = o
    assert o + 32 == len(b), "the layout consumes the blob to its last byte"
    return s


def damaged_blobs():
    """(name, fwd, bwd, expected detail)"""
    out = []
    pairs = [("golden", blob("fwd"), blob("rev"))]
    d = rc.create_dfa(r"from:[^\r\n]*<([a-z]+)@example\.com>\r\n")
    pairs.append(("compiled", d.fwd, d.bwd))
    for tag, fwd, bwd in pairs:
        out.append((f"{tag} intact", fwd, bwd, 0))
        for side, off in (("fwd", 0), ("bwd", A.D_DFA_BWD_OFFSET)):
            good = fwd if side == "fwd" else bwd
            st = section_starts(good)
            for sec, at in st.items():
                cut = good[:at + (1 if sec != A.D_DFA_LABEL else 5)]          # the blob ends inside this section
                f2, b2 = (cut, bwd) if side == "fwd" else (fwd, cut)
                out.append((f"{tag} {side} cut in section {sec}", f2, b2, sec + off))

            def put(at, val, f=good):
                return f[:at] + struct.pack("<I", val) + f[at + 4:]
            dmg = [
                ("label byte", good[:5] + b"X" + good[6:], A.D_DFA_LABEL),
                ("big-endian marker", put(32, 0xFFFE0000), A.D_DFA_ENDIAN_VERSION),
                ("version 3", put(36, 3), A.D_DFA_ENDIAN_VERSION),
                ("stride2 = 0", put(52, 0), A.D_DFA_TRANSITIONS),
                # a byte class beyond the alphabet: table[state + class] would land in columns nobody id-checked (found by
                # tests/cpp/host_fuzz.cpp; ByteClasses::from_bytes rejects it too)
                ("class beyond the alphabet", good[:56 + 65] + bytes([good[56 + 255] + 2]) + good[56 + 66:], A.D_DFA_TRANSITIONS),
                ("unaligned transition", put(48 + 8 + 256 + 4 * (1 << struct.unpack_from('<I', good, 52)[0]), 3), A.D_DFA_TRANSITIONS),
                ("start kind 7", put(st[A.D_DFA_START_TABLE], 7), A.D_DFA_START_TABLE),
                ("start stride 5", put(st[A.D_DFA_START_TABLE] + 4 + 256, 5), A.D_DFA_START_TABLE),
                ("match states > states", put(st[A.D_DFA_MATCH_STATES], 0x7FFFFFFF), A.D_DFA_MATCH_STATES),
                ("min_match > max_match", put(st[A.D_DFA_SPECIAL] + 8, 0x7FFFFF00), A.D_DFA_SPECIAL),
                ("accelerators > states", put(st[A.D_DFA_ACCELS], 0x7FFFFFFF), A.D_DFA_ACCELS),
            ]
            for name, bad, sec in dmg:
                f2, b2 = (bad, bwd) if side == "fwd" else (fwd, bad)
                out.append((f"{tag} {side} {name}", f2, b2, sec + off))
    out.append(("junk", b"junk", b"junk", A.D_DFA_LABEL))
    out.append(("empty", b"", b"", A.D_DFA_LABEL))
    out.append(("three flag words (the layout SURVEY A.3 recalled)", blob("fwd")[:44] + struct.pack("<III", 0, 1, 0) + blob("fwd")[48:], blob("rev"),
                None))      # fails somewhere behind the flags: which section depends on the bytes; only "fails" is asserted
    return out


def test_oracle_names_the_failing_section(oracle):
    for name, fwd, bwd, exp in damaged_blobs():
        got = oracle.dfa_status(oracle.dfa_register(fwd, bwd))
        if exp is None:
            assert got != 0, name
        else:
            assert got == exp, (name, got, exp)


@pytest.mark.gpu
def test_engine_names_the_failing_section_and_records_carry_it(engine, oracle):
    import synth
    inputs, wl, _ = synth.make_regex_workload("sections", 2, 500, n_header_parts=1, n_body_parts=0, seed=4)
    for name, fwd, bwd, exp in damaged_blobs():
        i1, i2 = engine.dfa_register(fwd, bwd), oracle.dfa_register(fwd, bwd)
        got = engine.dfa_status(i1)
        assert got == oracle.dfa_status(i2), name
        if exp is not None:
            assert got == exp, (name, got, exp)
        ins = [A.EmailWithRegex(i.email, A.RegexInfo([A.CompiledRegex(A.DFA(fwd, bwd), None)], None)) for i in inputs]
        rec = engine.verify_batch(engine.pack_with_regex(ins))
        orec = oracle.verify_batch(oracle.pack_with_regex(ins))
        assert (rec["status"] == orec["status"]).all() and (rec["detail"] == orec["detail"]).all(), name
        if got:
            assert all(int(s) == A.ZKE_DFA_DECODE_FAIL and int(d) == got for s, d in zip(rec["status"], rec["detail"])), name


@pytest.mark.gpu
def test_unregister_and_registry_cap(oracle):
    """zke_dfa_unregister frees an id (batches naming it then report ZKE_D_DFA_UNREGISTERED); a full registry (zke_options.max_dfas)
    evicts the pairs zke_verify_email_with_regex registered on its own, least recently used first, and refuses explicit
    registrations with a message instead of growing without bound."""
    import synth
    import zkemail_rs_amd as z
    eng = z.Engine(max_dfas=64)                      # (the floor of the option)
    try:
        inputs, wl, _ = synth.make_regex_workload("unreg", 2, 500, n_header_parts=1, n_body_parts=0, seed=6)
        packed = eng.pack_with_regex(inputs)
        rid = int(packed.hdr_ids[0])
        assert (eng.verify_batch(packed)["status"] == 0).all()
        eng.dfa_unregister(rid)
        rec = eng.verify_batch(packed)
        assert all((int(s), int(d)) == (A.ZKE_DFA_DECODE_FAIL, A.D_DFA_UNREGISTERED) for s, d in zip(rec["status"], rec["detail"]))
        with pytest.raises(z.EngineError):
            eng.dfa_unregister(rid)
        # per-e-mail entry: 80 distinct pairs through a registry of 64 — every call still verifies (the oldest are evicted)
        base = inputs[0]
        for k in range(80):
            d = rc.create_dfa(r"subject:([^\r\n]+)\r\n|zz%d" % k)
            one = A.EmailWithRegex(base.email, A.RegexInfo([A.CompiledRegex(d, None)], None))
            out = eng.verify_email_with_regex(one)
            assert out.email.public_key_hash
        # explicit registrations are never evicted: the 65th is refused
        eng2 = z.Engine(max_dfas=64)
        try:
            with pytest.raises(z.EngineError, match="registry full"):
                for k in range(70):
                    eng2.dfa_register(b"junk%d" % k, b"junk")
        finally:
            eng2.close()
    finally:
        eng.close()


@pytest.mark.gpu
def test_per_email_regex_calls_from_four_threads_through_a_full_registry():
    """96 distinct pairs through a registry of 64 from four threads at once: registrations evict each other's pairs all the
    time and ids change hands, yet every call must be checked against ITS pattern — the pairs of a call in progress are pinned
    (csrc/pipeline.hip.h, dfa_register_impl / dfa_unpin).  Pattern k matches one six-byte window of the subject and the
    call's capture is that window: under any other pattern of the set the capture is contained in no match (regex.rs:41-47)
    and the call panics."""
    import threading

    import synth
    import zkemail_rs_amd as z
    eng = z.Engine(max_dfas=64, slots=3)
    try:
        inputs, wl, _ = synth.make_regex_workload("pins", 2, 500, n_header_parts=1, n_body_parts=0, seed=6)
        base = inputs[0]
        subject = [ln for ln in base.email.raw_email.split(b"\r\n") if ln.lower().startswith(b"subject:")][0].split(b":", 1)[1].strip().decode()
        windows = [subject[k:k + 6] for k in range(32)]
        assert len(set(windows)) == 32 and all(subject.count(w) == 1 for w in windows)
        esc = lambda s: "".join(c if c.isalnum() else "\\x%02x" % ord(c) for c in s)
        ones, want = [], []
        for w in windows:
            for v in "ABC":
                ones.append(A.EmailWithRegex(base.email, A.RegexInfo([A.CompiledRegex(rc.create_dfa(esc(w) + "|zz" + v), [w])], None)))
                want.append([w])
        for one, m in zip(ones, want):                              # serially first: every call passes with its own pattern ...
            assert eng.verify_email_with_regex(one).regex_matches == m
        wrong = A.EmailWithRegex(base.email, A.RegexInfo([A.CompiledRegex(ones[0].regex_info.header_parts[0].verify_re, [windows[9]])], None))
        with pytest.raises(z.VerifyPanic):                          # ... and panics with another one's capture
            eng.verify_email_with_regex(wrong)
        errors = []

        def worker(tid):
            try:
                for it in range(150):
                    k = (tid * 29 + it * 7) % len(ones)
                    if eng.verify_email_with_regex(ones[k]).regex_matches != want[k]:
                        raise AssertionError(f"thread {tid} call {it}")
            except Exception as ex:
                errors.append(f"thread {tid}: {ex!r}")

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not errors, errors[:3]
        assert not any(t.is_alive() for t in ths)
    finally:
        eng.close()
