"""Regex -> dense DFA in regex-automata 0.4 little-endian wire format.

Host-side mirror of helpers/src/regex.rs: ``create_dfa`` (``to_bytes_little_endian`` with the
leading padding stripped, :7-14) and ``compile_regex_parts`` (:16-51).  The reference does this
with regex-automata's own compiler; no Rust toolchain exists here, so this is an independent
compiler that emits the SAME WIRE FORMAT with the SAME SEARCH SEMANTICS (leftmost-first forward
DFA with one-byte-delayed match states and an EOI transition; anchored, match-kind-all reverse
DFA), not byte-identical tables.  It is byte-oriented: ``.`` is ``[^\\n]`` over bytes, classes
are byte classes, no Unicode tables, no ``\\b``; flags.is_utf8 is 0 unless asked otherwise.

Supported syntax: literals, escapes (\\d \\w \\s \\D \\W \\S \\n \\r \\t \\f \\v \\0 \\xHH and escaped
punctuation), ``.``, classes ``[a-z0-9_]`` / ``[^...]``, groups ``( )`` ``(?: )``, alternation,
greedy and lazy ``* + ? {m} {m,} {m,n}``, ``^`` and ``$`` (text anchors, not multi-line).
"""
from __future__ import annotations

import re as _pyre
import struct
from dataclasses import dataclass
from typing import Dict, FrozenSet, List, Optional, Sequence, Tuple

from ._abi import CompiledRegex, DFA

LABEL = b"rust-regex-automata-dfa-dense"
DEAD = 0
NONE32 = 0xFFFFFFFF


class RegexSyntaxError(ValueError):
    pass


# ------------------------------------------------------------------ AST
@dataclass
class Node:
    kind: str                     # lit, cat, alt, rep, start, end, empty, group
    byteset: Optional[FrozenSet[int]] = None
    kids: Optional[List["Node"]] = None
    lo: int = 0
    hi: Optional[int] = None      # None = unbounded
    greedy: bool = True


_ESC_CLASS = {
    "d": frozenset(range(0x30, 0x3A)),
    "w": frozenset(list(range(0x30, 0x3A)) + list(range(0x41, 0x5B)) + list(range(0x61, 0x7B)) + [0x5F]),
    "s": frozenset([0x20, 0x09, 0x0A, 0x0B, 0x0C, 0x0D]),
}
_ALL = frozenset(range(256))
_ESC_CHAR = {"n": 0x0A, "r": 0x0D, "t": 0x09, "f": 0x0C, "v": 0x0B, "0": 0x00, "a": 0x07}


class _Parser:
    def __init__(self, pat: str):
        self.s = pat
        self.i = 0
        self.ngroups = 0

    def peek(self):
        return self.s[self.i] if self.i < len(self.s) else None

    def eat(self):
        c = self.s[self.i]
        self.i += 1
        return c

    def parse(self) -> Node:
        n = self.alt()
        if self.i != len(self.s):
            raise RegexSyntaxError(f"unexpected {self.s[self.i]!r} at {self.i}")
        return n

    def alt(self) -> Node:
        branches = [self.cat()]
        while self.peek() == "|":
            self.eat()
            branches.append(self.cat())
        return branches[0] if len(branches) == 1 else Node("alt", kids=branches)

    def cat(self) -> Node:
        items = []
        while self.peek() is not None and self.peek() not in "|)":
            items.append(self.rep())
        if not items:
            return Node("empty")
        return items[0] if len(items) == 1 else Node("cat", kids=items)

    def rep(self) -> Node:
        a = self.atom()
        while True:
            c = self.peek()
            if c == "*":
                self.eat(); lo, hi = 0, None
            elif c == "+":
                self.eat(); lo, hi = 1, None
            elif c == "?":
                self.eat(); lo, hi = 0, 1
            elif c == "{":
                m = _pyre.match(r"\{(\d+)(?:(,)(\d*))?\}", self.s[self.i:])
                if not m:
                    break
                self.i += m.end()
                lo = int(m.group(1))
                hi = lo if not m.group(2) else (int(m.group(3)) if m.group(3) else None)
                if hi is not None and hi < lo:
                    raise RegexSyntaxError("bad repetition range")
            else:
                break
            greedy = True
            if self.peek() == "?":
                self.eat(); greedy = False
            if a.kind in ("start", "end"):
                raise RegexSyntaxError("repetition of an anchor")
            a = Node("rep", kids=[a], lo=lo, hi=hi, greedy=greedy)
        return a

    def escape(self, in_class: bool):
        if self.peek() is None:
            raise RegexSyntaxError("dangling backslash")
        c = self.eat()
        if c in "dws":
            return _ESC_CLASS[c]
        if c in "DWS":
            return _ALL - _ESC_CLASS[c.lower()]
        if c == "x":
            h = self.s[self.i:self.i + 2]
            if len(h) != 2:
                raise RegexSyntaxError("bad \\x escape")
            self.i += 2
            return frozenset([int(h, 16)])
        if c in _ESC_CHAR:
            return frozenset([_ESC_CHAR[c]])
        if c.isalnum():
            raise RegexSyntaxError(f"unsupported escape \\{c}")
        return frozenset(c.encode("utf-8")) if len(c.encode("utf-8")) == 1 else None

    def atom(self) -> Node:
        c = self.eat()
        if c == "(":
            if self.s.startswith("?:", self.i):
                self.i += 2
            elif self.peek() == "?":
                raise RegexSyntaxError("unsupported group flag")
            else:
                self.ngroups += 1
            n = self.alt()
            if self.peek() != ")":
                raise RegexSyntaxError("missing )")
            self.eat()
            return Node("group", kids=[n])
        if c == "[":
            return self.cls()
        if c == ".":
            return Node("lit", byteset=_ALL - {0x0A})
        if c == "^":
            return Node("start")
        if c == "$":
            return Node("end")
        if c == "\\":
            bs = self.escape(False)
            if bs is None:
                raise RegexSyntaxError("non-ASCII escape")
            return Node("lit", byteset=bs)
        if c in "*+?{":
            if c == "{":
                return Node("lit", byteset=frozenset([0x7B]))
            raise RegexSyntaxError(f"nothing to repeat at {self.i - 1}")
        b = c.encode("utf-8")
        if len(b) == 1:
            return Node("lit", byteset=frozenset(b))
        return Node("cat", kids=[Node("lit", byteset=frozenset([x])) for x in b])   # a UTF-8 literal is its bytes

    def cls(self) -> Node:
        neg = False
        if self.peek() == "^":
            self.eat(); neg = True
        items: set = set()
        first = True
        while True:
            if self.peek() is None:
                raise RegexSyntaxError("missing ]")
            c = self.eat()
            if c == "]" and not first:
                break
            first = False
            if c == "\\":
                bs = self.escape(True)
                if bs is None:
                    raise RegexSyntaxError("non-ASCII in class")
                lo_set = bs
            else:
                b = c.encode("utf-8")
                if len(b) != 1:
                    raise RegexSyntaxError("non-ASCII in class (byte-oriented compiler)")
                lo_set = frozenset(b)
            if len(lo_set) == 1 and self.peek() == "-" and self.i + 1 < len(self.s) and self.s[self.i + 1] != "]":
                self.eat()
                d = self.eat()
                if d == "\\":
                    hs = self.escape(True)
                    if hs is None or len(hs) != 1:
                        raise RegexSyntaxError("bad range end")
                    hi = next(iter(hs))
                else:
                    hb = d.encode("utf-8")
                    if len(hb) != 1:
                        raise RegexSyntaxError("non-ASCII in class")
                    hi = hb[0]
                lo = next(iter(lo_set))
                if hi < lo:
                    raise RegexSyntaxError("reversed range")
                items.update(range(lo, hi + 1))
            else:
                items.update(lo_set)
        bs = frozenset(items)
        return Node("lit", byteset=(_ALL - bs) if neg else bs)


def _reverse(n: Node) -> Node:
    if n.kind == "cat":
        return Node("cat", kids=[_reverse(k) for k in reversed(n.kids)])
    if n.kind in ("alt", "group"):
        return Node(n.kind, kids=[_reverse(k) for k in n.kids])
    if n.kind == "rep":
        return Node("rep", kids=[_reverse(n.kids[0])], lo=n.lo, hi=n.hi, greedy=n.greedy)
    return n   # lit / empty / start / end keep their meaning (start = start of haystack)


def _always_start_anchored(n: Node) -> bool:
    if n.kind == "start":
        return True
    if n.kind == "group":
        return _always_start_anchored(n.kids[0])
    if n.kind == "cat":
        return _always_start_anchored(n.kids[0])
    if n.kind == "alt":
        return all(_always_start_anchored(k) for k in n.kids)
    return False


# ------------------------------------------------------------------ Thompson NFA (priority ordered)
class _NFA:
    # state kinds: ("byte", set, nxt) ("split", [nxt...]) ("start", nxt) ("end", nxt) ("match",)
    def __init__(self):
        self.st: List[tuple] = []

    def add(self, t) -> int:
        self.st.append(t)
        return len(self.st) - 1

    def patch(self, i, t):
        self.st[i] = t

    def build(self, n: Node, nxt: int) -> int:
        """Return the entry state of `n` continuing to `nxt`."""
        k = n.kind
        if k == "empty":
            return nxt
        if k == "lit":
            return self.add(("byte", n.byteset, nxt))
        if k == "group":
            return self.build(n.kids[0], nxt)
        if k == "cat":
            for kid in reversed(n.kids):
                nxt = self.build(kid, nxt)
            return nxt
        if k == "alt":
            return self.add(("split", [self.build(kid, nxt) for kid in n.kids]))
        if k == "start":
            return self.add(("start", nxt))
        if k == "end":
            return self.add(("end", nxt))
        if k == "rep":
            kid, lo, hi, greedy = n.kids[0], n.lo, n.hi, n.greedy
            if hi is None:
                loop = self.add(("split", []))
                body = self.build(kid, loop)
                self.patch(loop, ("split", [body, nxt] if greedy else [nxt, body]))
                entry = loop
            else:
                entry = nxt
                for _ in range(hi - lo):
                    body = self.build(kid, entry)
                    entry = self.add(("split", [body, nxt] if greedy else [nxt, body]))
            for _ in range(lo):
                entry = self.build(kid, entry)
            return entry
        raise AssertionError(k)


def _closure(nfa: _NFA, seeds: Sequence[int], look_start: bool, look_end: bool) -> Tuple[int, ...]:
    """Priority-ordered epsilon closure.  Assertion states whose look is not (yet) satisfied stay in the set."""
    out: List[int] = []
    seen = set()
    stack = list(reversed(seeds))
    while stack:
        s = stack.pop()
        if s in seen:
            continue
        seen.add(s)
        t = nfa.st[s]
        if t[0] == "split":
            stack.extend(reversed(t[1]))
        elif t[0] == "start":
            if look_start:
                stack.append(t[1])
            else:
                out.append(s)
        elif t[0] == "end":
            if look_end:
                stack.append(t[1])
            else:
                out.append(s)
        else:
            out.append(s)
    return tuple(out)


def _can_match_empty(n: Node) -> bool:
    k = n.kind
    if k in ("empty", "start", "end"):
        return True
    if k == "lit":
        return False
    if k == "group":
        return _can_match_empty(n.kids[0])
    if k == "cat":
        return all(_can_match_empty(x) for x in n.kids)
    if k == "alt":
        return any(_can_match_empty(x) for x in n.kids)
    if k == "rep":
        return n.lo == 0 or _can_match_empty(n.kids[0])
    raise AssertionError(k)


def _byte_classes(nfa: _NFA) -> List[int]:
    """regex-automata ByteClassSet: contiguous byte ranges; a new class starts wherever some byte set of
    the NFA changes membership, so classes[255] is the largest id and alphabet_len = classes[255] + 2."""
    boundary = [False] * 256
    for t in nfa.st:
        if t[0] == "byte":
            bs = t[1]
            for b in range(1, 256):
                if (b in bs) != ((b - 1) in bs):
                    boundary[b] = True
    classes, c = [0] * 256, 0
    for b in range(256):
        if boundary[b]:
            c += 1
        classes[b] = c
    return classes


@dataclass
class _Built:
    table: List[List[int]]          # [state][class] -> state index (incl. the EOI column)
    is_match: List[bool]
    starts: List[int]               # 12 state indices (unanchored[6], anchored[6])
    classes: List[int]
    alphabet_len: int


def _determinize(nfa: _NFA, start: int, classes: List[int], *, reverse: bool, leftmost_first: bool,
                 want_unanchored: bool) -> _Built:
    ncls = max(classes) + 1
    alphabet_len = ncls + 1
    rep_byte = [None] * ncls
    for b in range(256):
        if rep_byte[classes[b]] is None:
            rep_byte[classes[b]] = b
    # unanchored prefix: a lowest-priority any-byte loop in front of the pattern ((?s-u:.)*?)
    un_start = None
    if want_unanchored:
        loop = nfa.add(("split", []))
        anyb = nfa.add(("byte", _ALL, loop))
        nfa.patch(loop, ("split", [start, anyb]))
        un_start = loop

    index: Dict[Tuple[Tuple[int, ...], bool], int] = {}
    states: List[Tuple[Tuple[int, ...], bool]] = []
    table: List[List[int]] = []

    def intern(key) -> int:
        if key == ((), False):
            return DEAD
        if key not in index:
            index[key] = len(states)
            states.append(key)
            table.append([DEAD] * alphabet_len)
        return index[key]

    states.append(((), False)); table.append([DEAD] * alphabet_len)   # state 0 = dead
    index[((), False)] = 0

    def start_state(seed: int, text: bool) -> int:
        # forward: Start::Text satisfies ^ ; reverse: Start::Text means "at the end of the haystack" and satisfies $
        clo = _closure(nfa, [seed], look_start=(text and not reverse), look_end=(text and reverse))
        return intern((clo, False))

    starts = []
    for seed in ([un_start] if want_unanchored else [None]) + [start]:
        for cfg in range(6):
            if seed is None:
                starts.append(DEAD)
            else:
                starts.append(start_state(seed, cfg == 2))

    done = 0
    while done < len(states):
        nfa_set, _flag = states[done]
        si = done
        done += 1
        if si == DEAD:
            continue

        def step(byte: Optional[int]) -> Tuple[Tuple[int, ...], bool]:
            cur = nfa_set
            if byte is None:   # EOI: the haystack edge satisfies $ going forward, ^ going backward
                cur = _closure(nfa, list(cur), look_start=reverse, look_end=not reverse)
            nxt: List[int] = []
            matched = False
            for s in cur:
                t = nfa.st[s]
                if t[0] == "match":
                    matched = True
                    if leftmost_first:
                        break          # lower-priority threads are cut
                elif t[0] == "byte" and byte is not None and byte in t[1]:
                    nxt.append(t[2])
            clo = _closure(nfa, nxt, False, False) if nxt else ()
            return (clo, matched)

        for c in range(ncls):
            table[si][c] = intern(step(rep_byte[c]))
        table[si][alphabet_len - 1] = intern(step(None))
    return _Built(table, [f for (_s, f) in states], starts, classes, alphabet_len)


def _serialize(b: _Built, *, start_kind: int, has_empty: bool, is_utf8: bool, always_anchored: bool) -> bytes:
    n = len(b.table)
    # layout: dead, match states, everything else (Special: only dead + match states are special)
    order = [0] + [i for i in range(1, n) if b.is_match[i]] + [i for i in range(1, n) if not b.is_match[i]]
    newidx = {old: new for new, old in enumerate(order)}
    stride2 = max(1, (b.alphabet_len - 1).bit_length())
    stride = 1 << stride2
    sid = lambda i: newidx[i] << stride2   # noqa: E731
    nm = sum(1 for i in range(1, n) if b.is_match[i])
    out = bytearray()
    out += LABEL + b"\0" * (32 - len(LABEL))
    out += struct.pack("<III", 0xFEFF, 2, 0)
    out += struct.pack("<III", int(has_empty), int(is_utf8), int(always_anchored))
    out += struct.pack("<II", n, stride2)
    out += bytes(b.classes)
    tbl = [0] * (n * stride)
    for old in range(n):
        base = newidx[old] * stride
        for c in range(b.alphabet_len):
            tbl[base + c] = sid(b.table[old][c])
    out += struct.pack(f"<{len(tbl)}I", *tbl)
    # start table
    start_map = bytearray(256)   # Start::NonWordByte = 0
    for ch in range(256):
        if ch == 0x0A:
            start_map[ch] = 3
        elif ch == 0x0D:
            start_map[ch] = 4
        elif ch == 0x5F or 0x30 <= ch <= 0x39 or 0x41 <= ch <= 0x5A or 0x61 <= ch <= 0x7A:
            start_map[ch] = 1
    st_ids = [sid(s) for s in b.starts]
    uni_un = st_ids[0] if (start_kind != 2 and len(set(st_ids[:6])) == 1) else NONE32
    uni_an = st_ids[6] if (start_kind != 1 and len(set(st_ids[6:])) == 1) else NONE32
    out += struct.pack("<I", start_kind) + bytes(start_map) + struct.pack("<IIII", 6, NONE32, uni_un, uni_an)
    out += struct.pack("<12I", *st_ids)
    # match states: (start, len) pairs into pattern_ids, one pattern (id 0)
    out += struct.pack("<I", nm)
    for k in range(nm):
        out += struct.pack("<II", k, 1)
    out += struct.pack("<II", 1, nm) + struct.pack(f"<{nm}I", *([0] * nm))
    # special
    min_match = (1 << stride2) if nm else DEAD
    max_match = (nm << stride2) if nm else DEAD
    out += struct.pack("<8I", max_match, DEAD, min_match, max_match, DEAD, DEAD, DEAD, DEAD)
    out += struct.pack("<I", 0)          # accelerators: none
    out += b"\0" * 32                    # quit set: empty
    return bytes(out)


def create_dfa(pattern: str, *, is_utf8: bool = False) -> DFA:
    """helpers/src/regex.rs:7-14 create_dfa: forward + reverse dense DFA blobs, padding stripped."""
    ast = _Parser(pattern).parse()
    has_empty = _can_match_empty(ast)
    anchored = _always_start_anchored(ast)
    # forward: leftmost-first, StartKind::Both
    nf = _NFA()
    m = nf.add(("match",))
    s = nf.build(ast, m)
    classes = _byte_classes(nf)
    fwd = _determinize(nf, s, classes, reverse=False, leftmost_first=True, want_unanchored=not anchored)
    if anchored:   # the unanchored start of an always-anchored NFA is its anchored start
        fwd.starts[:6] = fwd.starts[6:]
    # reverse: anchored only, MatchKind::All (dfa::regex::Builder::build_many)
    nr = _NFA()
    mr = nr.add(("match",))
    sr = nr.build(_reverse(ast), mr)
    rclasses = _byte_classes(nr)
    rev = _determinize(nr, sr, rclasses, reverse=True, leftmost_first=False, want_unanchored=False)
    return DFA(
        fwd=_serialize(fwd, start_kind=0, has_empty=has_empty, is_utf8=is_utf8, always_anchored=anchored),
        bwd=_serialize(rev, start_kind=2, has_empty=has_empty, is_utf8=is_utf8, always_anchored=False),
    )


@dataclass
class RegexPattern:                    # helpers/src/structs.rs:3-7
    pattern: str
    capture_indices: Optional[List[int]] = None


@dataclass
class RegexConfig:                     # helpers/src/structs.rs:9-13 (the regex_config.json schema)
    header_parts: Optional[List[RegexPattern]] = None
    body_parts: Optional[List[RegexPattern]] = None

    @staticmethod
    def from_json(obj: dict) -> "RegexConfig":
        def parts(x):
            return None if x is None else [RegexPattern(p["pattern"], p.get("capture_indices")) for p in x]
        return RegexConfig(parts(obj.get("header_parts")), parts(obj.get("body_parts")))


def compile_regex_parts(parts: Sequence[RegexPattern], inp: bytes) -> List[CompiledRegex]:
    """helpers/src/regex.rs:16-51: one match exactly, capture strings by group index (lossy UTF-8)."""
    out = []
    for part in parts:
        rx = _pyre.compile(part.pattern.encode("utf-8"), _pyre.DOTALL if False else 0)
        ms = list(rx.finditer(inp))
        if len(ms) != 1:
            raise ValueError(f"Input doesn't match regex pattern: {part!r}")
        caps: List[str] = []
        for gi in part.capture_indices or []:
            if gi > rx.groups or ms[0].group(gi) is None:
                raise ValueError("Capture group not found")
            caps.append(ms[0].group(gi).decode("utf-8", errors="replace"))
        out.append(CompiledRegex(create_dfa(part.pattern), caps))
    return out
