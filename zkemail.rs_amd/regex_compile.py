"""Regex -> dense DFA in regex-automata 0.4 little-endian wire format.

Host-side mirror of helpers/src/regex.rs: ``create_dfa`` (``to_bytes_little_endian`` with the
leading padding stripped, :7-14) and ``compile_regex_parts`` (:16-51).  The reference does this
with regex-automata's own compiler; no Rust toolchain exists here, so this is an independent
compiler that emits the SAME WIRE FORMAT with the SAME SEARCH SEMANTICS (leftmost-first forward
DFA with one-byte-delayed match states and an EOI transition; anchored, match-kind-all reverse
DFA), not byte-identical tables.  Two modes: ``unicode=True`` is regex-automata's default syntax, the
one the reference compiles with — ``.``, classes, ``\\d \\w \\s`` and negations range over Unicode scalar
values and become UTF-8 byte-sequence automata, flags.is_utf8 = 1; ``unicode=False`` is ``(?-u)``
throughout: bytes, flags.is_utf8 = 0.  Look-around: ``^ $ \\A \\z``, ``(?m)`` line anchors (LF), the ASCII word boundary
``(?-u:\\b)`` / ``(?-u:\\B)``; a Unicode ``\\b`` is an error, as it is for the reference (a dense DFA cannot hold one).

Supported syntax: literals, escapes (\\d \\w \\s \\D \\W \\S \\n \\r \\t \\f \\v \\0 \\xHH and escaped
punctuation), ``.``, classes ``[a-z0-9_]`` / ``[^...]``, groups ``( )`` ``(?: )``, alternation,
greedy and lazy ``* + ? {m} {m,} {m,n}``, ``^`` ``$`` ``\\A`` ``\\z`` (text anchors, not multi-line), POSIX bracket classes
``[[:alpha:]]`` ``[[:^digit:]]``, ``\\x{HHHH}``, non-ASCII
literals and class members, the inline flags ``(?s) (?-s) (?u) (?-u) (?i) (?-i)`` and their scoped forms
(``(?i)``: regex-syntax's simple case folding — Unicode orbits in Unicode mode, ASCII letters in byte mode).
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
    kind: str                     # lit, cat, alt, rep, empty, group; assertions: start, end, sol, eol, wb, nwb
    byteset: Optional[FrozenSet[int]] = None
    kids: Optional[List["Node"]] = None
    lo: int = 0
    hi: Optional[int] = None      # None = unbounded
    greedy: bool = True


_ALL = frozenset(range(256))
_ESC_CHAR = {"n": 0x0A, "r": 0x0D, "t": 0x09, "f": 0x0C, "v": 0x0B, "0": 0x00, "a": 0x07}
MAXCP = 0x10FFFF
Ranges = List[Tuple[int, int]]          # sorted, disjoint, inclusive ranges of bytes (byte mode) or code points (Unicode mode)

_ASCII_CLASS: Dict[str, Ranges] = {
    "d": [(0x30, 0x39)],
    "w": [(0x30, 0x39), (0x41, 0x5A), (0x5F, 0x5F), (0x61, 0x7A)],
    "s": [(0x09, 0x0D), (0x20, 0x20)],
}
# White_Space (PropList.txt) — what regex-syntax's Unicode \s is
_UNI_SPACE: Ranges = [(0x09, 0x0D), (0x20, 0x20), (0x85, 0x85), (0xA0, 0xA0), (0x1680, 0x1680), (0x2000, 0x200A), (0x2028, 0x2029),
                      (0x202F, 0x202F), (0x205F, 0x205F), (0x3000, 0x3000)]
_POSIX_CLASS: Dict[str, Ranges] = {
    "alnum": [(0x30, 0x39), (0x41, 0x5A), (0x61, 0x7A)], "alpha": [(0x41, 0x5A), (0x61, 0x7A)], "ascii": [(0x00, 0x7F)],
    "blank": [(0x09, 0x09), (0x20, 0x20)], "cntrl": [(0x00, 0x1F), (0x7F, 0x7F)], "digit": [(0x30, 0x39)], "graph": [(0x21, 0x7E)],
    "lower": [(0x61, 0x7A)], "print": [(0x20, 0x7E)], "punct": [(0x21, 0x2F), (0x3A, 0x40), (0x5B, 0x60), (0x7B, 0x7E)],
    "space": [(0x09, 0x0D), (0x20, 0x20)], "upper": [(0x41, 0x5A)], "word": [(0x30, 0x39), (0x41, 0x5A), (0x5F, 0x5F), (0x61, 0x7A)],
    "xdigit": [(0x30, 0x39), (0x41, 0x46), (0x61, 0x66)],
}
_UNI_CACHE: Dict[str, Ranges] = {}


def _norm(r: Sequence[Tuple[int, int]]) -> Ranges:
    out: Ranges = []
    for lo, hi in sorted(r):
        if out and lo <= out[-1][1] + 1:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def _negate(r: Ranges, top: int) -> Ranges:
    out, nxt = [], 0
    for lo, hi in _norm(r):
        if lo > nxt:
            out.append((nxt, lo - 1))
        nxt = hi + 1
    if nxt <= top:
        out.append((nxt, top))
    return out


_FOLD_ORBITS: Optional[List[Tuple[int, ...]]] = None


def _fold_orbits() -> List[Tuple[int, ...]]:
    """Simple case folding (CaseFolding.txt statuses C + S, what regex-syntax folds with) as orbits of more than one scalar
    value.  Python has no table of it: `str.casefold` is C + F, so a one-character result is the C mapping; where the full
    folding is several characters the S mapping, when there is one, is the one-character `lower()` (U+1E9E -> U+00DF, the
    Greek capitals with ypogegrammeni); U+0130 has neither and stays alone, U+0131 folds to itself."""
    global _FOLD_ORBITS
    if _FOLD_ORBITS is None:
        by_key: Dict[int, List[int]] = {}
        for cp in range(MAXCP + 1):
            if 0xD800 <= cp <= 0xDFFF:
                continue
            ch = chr(cp)
            f = ch.casefold()
            if len(f) != 1:
                f = ch.lower()
                if len(f) != 1:
                    f = ch
            k = ord(f)
            if k != cp:
                by_key.setdefault(k, [k]).append(cp)
        _FOLD_ORBITS = [tuple(sorted(set(v))) for v in by_key.values()]
    return _FOLD_ORBITS


def _case_fold(r: Ranges, unicode: bool) -> Ranges:
    """The class closed under simple case folding ((?i); regex-syntax ClassUnicode / ClassBytes::case_fold_simple)."""
    import bisect
    r = _norm(r)
    los = [lo for lo, _ in r]

    def has(cp: int) -> bool:
        k = bisect.bisect_right(los, cp) - 1
        return k >= 0 and r[k][1] >= cp
    extra: Ranges = []
    if unicode:
        for orbit in _fold_orbits():
            if any(has(c) for c in orbit):
                extra += [(c, c) for c in orbit]
    else:
        for c in range(26):
            if has(0x41 + c) or has(0x61 + c):
                extra += [(0x41 + c, 0x41 + c), (0x61 + c, 0x61 + c)]
    return _norm(list(r) + extra)


def _unicode_class(name: str) -> Ranges:
    """\\d = Nd, \\w = Alphabetic + M + Nd + Pc + Join_Control, \\s = White_Space (regex-syntax, Unicode mode; UTS #18 Annex C).
    Tables come from the `regex` module's Unicode database when it is importable (one scan of the code space, cached),
    else from `unicodedata` general categories (no Other_Alphabetic: a few hundred combining vowel signs short)."""
    if name == "s":
        return _UNI_SPACE
    if name in _UNI_CACHE:
        return _UNI_CACHE[name]
    try:
        import regex as _rx
        rx = _rx.compile(r"\p{Nd}" if name == "d" else r"[\p{Alphabetic}\p{M}\p{Nd}\p{Pc}\p{Join_Control}]")
        hit = lambda cp: rx.match(chr(cp)) is not None   # noqa: E731
    except ImportError:
        import unicodedata as _ud
        cats = {"Nd"} if name == "d" else {"Lu", "Ll", "Lt", "Lm", "Lo", "Nl", "Mn", "Mc", "Me", "Nd", "Pc"}
        hit = lambda cp: _ud.category(chr(cp)) in cats or (name == "w" and cp in (0x200C, 0x200D))   # noqa: E731
    out: Ranges = []
    start = None
    for cp in range(MAXCP + 2):
        ok = cp <= MAXCP and not (0xD800 <= cp <= 0xDFFF) and hit(cp)
        if ok and start is None:
            start = cp
        elif not ok and start is not None:
            out.append((start, cp - 1))
            start = None
    _UNI_CACHE[name] = out
    return out


def _utf8_sequences(lo: int, hi: int) -> List[List[Tuple[int, int]]]:
    """The scalar-value range [lo, hi] as UTF-8 byte-range sequences (each sequence: one byte range per position), in
    ascending order — the utf8-ranges construction regex-syntax uses (RFC 3629; surrogates are not scalar values)."""
    out: List[List[Tuple[int, int]]] = []
    stack = [(lo, hi)]
    while stack:
        lo, hi = stack.pop()
        if lo > hi:
            continue
        if lo <= 0xDFFF and hi >= 0xD800:                    # cut the surrogate gap out
            stack.append((0xE000, hi))
            stack.append((lo, 0xD7FF))
            continue
        split = False
        for mx in (0x7F, 0x7FF, 0xFFFF):                     # one encoded length per piece
            if lo <= mx < hi:
                stack.append((mx + 1, hi))
                stack.append((lo, mx))
                split = True
                break
        if split:
            continue
        if hi <= 0x7F:
            out.append([(lo, hi)])
            continue
        for i in (1, 2, 3):                                   # continuation bytes must span their whole range
            m = (1 << (6 * i)) - 1
            if (lo & ~m) != (hi & ~m):
                if (lo & m) != 0:
                    stack.append(((lo | m) + 1, hi))
                    stack.append((lo, lo | m))
                    split = True
                    break
                if (hi & m) != m:
                    stack.append((hi & ~m, hi))
                    stack.append((lo, (hi & ~m) - 1))
                    split = True
                    break
        if split:
            continue
        a, b = chr(lo).encode("utf-8"), chr(hi).encode("utf-8")
        out.append(list(zip(a, b)))
    return out


def _bytes_node(lo: int, hi: int) -> Node:
    return Node("lit", byteset=frozenset(range(lo, hi + 1)))


def _seq_trie_node(seqs: List[List[Tuple[int, int]]]) -> Node:
    """Alternation of byte-range sequences with common leading ranges shared (keeps the NFA, and so the subset construction,
    small: a Unicode \\w is ~750 ranges)."""
    groups: Dict[Tuple[int, int], List[List[Tuple[int, int]]]] = {}
    order: List[Tuple[int, int]] = []
    for sq in seqs:
        if sq[0] not in groups:
            groups[sq[0]] = []
            order.append(sq[0])
        groups[sq[0]].append(sq[1:])
    alts = []
    for first in order:
        rests = [r for r in groups[first] if r]
        head = _bytes_node(*first)
        alts.append(head if not rests else Node("cat", kids=[head, _seq_trie_node(rests)]))
    return alts[0] if len(alts) == 1 else Node("alt", kids=alts)


def _class_node(r: Ranges, unicode: bool) -> Node:
    """A character class as an AST over BYTES: byte mode = one byte set; Unicode mode = the UTF-8 encodings of its scalar values."""
    r = _norm(r)
    if not unicode:
        return Node("lit", byteset=frozenset(b for lo, hi in r for b in range(lo, hi + 1)))
    if not r:
        return Node("lit", byteset=frozenset())               # matches nothing
    if r[-1][1] <= 0x7F:
        return Node("lit", byteset=frozenset(b for lo, hi in r for b in range(lo, hi + 1)))
    seqs: List[List[Tuple[int, int]]] = []
    ascii_bytes = set()
    for lo, hi in r:
        for sq in _utf8_sequences(lo, hi):
            if len(sq) == 1:
                ascii_bytes.update(range(sq[0][0], sq[0][1] + 1))
            else:
                seqs.append(sq)
    kids = []
    if ascii_bytes:
        kids.append(Node("lit", byteset=frozenset(ascii_bytes)))
    if seqs:
        t = _seq_trie_node(seqs)
        kids += t.kids if t.kind == "alt" else [t]
    return kids[0] if len(kids) == 1 else Node("alt", kids=kids)


class _Parser:
    """regex-syntax's surface, the part e-mail patterns use.  unicode=True (regex-automata's default, what
    helpers/src/regex.rs:20 builds with): `.`, classes, \\d \\w \\s and their negations range over Unicode scalar values and
    compile to UTF-8 byte sequences — a negated class never matches a stray byte >= 0x80.  unicode=False: bytes.
    Inline flags: (?s) (?u) (?i) (?m) and their negations, and the scoped forms (?s:...) (?-u:...) (?i:...) (?m:...)."""

    def __init__(self, pat: str, unicode: bool = False):
        self.s = pat
        self.i = 0
        self.ngroups = 0
        self.unicode = unicode
        self.dotall = False
        self.icase = False
        self.multiline = False

    @property
    def top(self) -> int:
        return MAXCP if self.unicode else 0xFF

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
        saved = (self.unicode, self.dotall, self.icase, self.multiline)       # (?flags) lasts to the end of the enclosing group
        branches = [self.cat()]
        while self.peek() == "|":
            self.eat()
            branches.append(self.cat())
        self.unicode, self.dotall, self.icase, self.multiline = saved
        return branches[0] if len(branches) == 1 else Node("alt", kids=branches)

    def cat(self) -> Node:
        items = []
        while self.peek() is not None and self.peek() not in "|)":
            a = self.rep()
            if a is not None:
                items.append(a)
        if not items:
            return Node("empty")
        return items[0] if len(items) == 1 else Node("cat", kids=items)

    def rep(self) -> Optional[Node]:
        a = self.atom()
        if a is None:
            return None
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
            if a.kind in ("start", "end", "sol", "eol", "wb", "nwb"):
                raise RegexSyntaxError("repetition of an assertion")
            a = Node("rep", kids=[a], lo=lo, hi=hi, greedy=greedy)
        return a

    def escape(self) -> Ranges:
        if self.peek() is None:
            raise RegexSyntaxError("dangling backslash")
        c = self.eat()
        if c in "dws":
            return _unicode_class(c) if self.unicode else _ASCII_CLASS[c]
        if c in "DWS":
            return _negate(_unicode_class(c.lower()) if self.unicode else _ASCII_CLASS[c.lower()], self.top)
        if c == "x":
            if self.peek() == "{":
                m = _pyre.match(r"\{([0-9a-fA-F]+)\}", self.s[self.i:])
                if not m:
                    raise RegexSyntaxError("bad \\x{...} escape")
                self.i += m.end()
                v = int(m.group(1), 16)
            else:
                h = self.s[self.i:self.i + 2]
                if len(h) != 2:
                    raise RegexSyntaxError("bad \\x escape")
                self.i += 2
                v = int(h, 16)
            if v > self.top or (self.unicode and 0xD800 <= v <= 0xDFFF):
                raise RegexSyntaxError("escape out of range")
            if self.unicode is False and v > 0xFF:
                raise RegexSyntaxError("escape out of range")
            return [(v, v)]
        if c in _ESC_CHAR:
            return [(_ESC_CHAR[c], _ESC_CHAR[c])]
        if c in "bB":
            raise RegexSyntaxError("word boundary inside a class")
        if c.isalnum():
            raise RegexSyntaxError(f"unsupported escape \\{c}")
        return [(ord(c), ord(c))] if (self.unicode or ord(c) < 0x80) else None

    def flags(self, text: str):
        on = True
        for ch in text:
            if ch == "-":
                on = False
            elif ch == "u":
                self.unicode = on
            elif ch == "s":
                self.dotall = on
            elif ch == "i":
                self.icase = on
            elif ch == "m":
                self.multiline = on
            else:
                raise RegexSyntaxError(f"unsupported flag {ch!r}")

    def literal(self, c: str) -> Node:
        if self.icase and (self.unicode or ord(c) < 0x80):
            folded = _case_fold([(ord(c), ord(c))], self.unicode)
            if folded != [(ord(c), ord(c))]:
                return _class_node(folded, self.unicode)
        b = c.encode("utf-8")
        if len(b) == 1:
            return Node("lit", byteset=frozenset(b))
        return Node("cat", kids=[Node("lit", byteset=frozenset([x])) for x in b])   # a non-ASCII literal is its UTF-8 bytes

    def atom(self) -> Optional[Node]:
        c = self.eat()
        if c == "(":
            if self.peek() == "?":
                m = _pyre.match(r"\?([a-z-]*)([:)])", self.s[self.i:])
                if not m:
                    raise RegexSyntaxError("unsupported group syntax")
                self.i += m.end()
                if m.group(2) == ")":                         # (?flags): applies to the rest of the enclosing group
                    self.flags(m.group(1))
                    return None
                saved = (self.unicode, self.dotall, self.icase, self.multiline)
                self.flags(m.group(1))                        # (?flags:...) and (?:...)
                n = self.alt()
                self.unicode, self.dotall, self.icase, self.multiline = saved
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
            return _class_node([(0, self.top)] if self.dotall else _negate([(0x0A, 0x0A)], self.top), self.unicode)
        if c == "^":
            return Node("sol" if self.multiline else "start")
        if c == "$":
            return Node("eol" if self.multiline else "end")
        if c == "\\":
            if self.peek() == "A":                            # \A, \z: the text anchors by their other names
                self.eat()
                return Node("start")
            if self.peek() == "z":
                self.eat()
                return Node("end")
            if self.peek() in ("b", "B"):
                # The Unicode word boundary cannot be built into a dense DFA: dense::Builder fails on it, so the reference's
                # compile_regex_parts (helpers/src/regex.rs:20) returns Err for such a pattern.  (?-u:\b) is the ASCII one.
                if self.unicode:
                    raise RegexSyntaxError("Unicode word boundary: regex-automata cannot build a dense DFA with it either; write (?-u:\\b)")
                return Node("wb" if self.eat() == "b" else "nwb")
            r = self.escape()
            if r is None:
                raise RegexSyntaxError("non-ASCII escape")
            return _class_node(_case_fold(r, self.unicode) if self.icase else r, self.unicode)
        if c in "*+?{":
            if c == "{":
                return Node("lit", byteset=frozenset([0x7B]))
            raise RegexSyntaxError(f"nothing to repeat at {self.i - 1}")
        return self.literal(c)

    def cls(self) -> Node:
        neg = False
        if self.peek() == "^":
            self.eat(); neg = True
        items: Ranges = []
        first = True

        def one(ch: str) -> int:
            if not self.unicode and ord(ch) > 0x7F:
                raise RegexSyntaxError("non-ASCII in a class of a byte-mode ((?-u)) pattern")
            return ord(ch)
        while True:
            if self.peek() is None:
                raise RegexSyntaxError("missing ]")
            c = self.eat()
            if c == "]" and not first:
                break
            first = False
            if c == "[" and self.peek() == ":":                # [[:alpha:]] ... — ASCII classes in both modes (regex-syntax)
                m = _pyre.match(r":(\^?)([a-z]+):\]", self.s[self.i:])
                if not m or m.group(2) not in _POSIX_CLASS:
                    raise RegexSyntaxError("unknown POSIX class")
                self.i += m.end()
                pr = _POSIX_CLASS[m.group(2)]
                items += _negate(pr, self.top) if m.group(1) else pr
                continue
            if c == "\\":
                lo_set = self.escape()
                if lo_set is None:
                    raise RegexSyntaxError("non-ASCII in class")
            else:
                lo_set = [(one(c), one(c))]
            single = len(lo_set) == 1 and lo_set[0][0] == lo_set[0][1]
            if single and self.peek() == "-" and self.i + 1 < len(self.s) and self.s[self.i + 1] != "]":
                self.eat()
                d = self.eat()
                if d == "\\":
                    hs = self.escape()
                    if hs is None or len(hs) != 1 or hs[0][0] != hs[0][1]:
                        raise RegexSyntaxError("bad range end")
                    hi = hs[0][0]
                else:
                    hi = one(d)
                lo = lo_set[0][0]
                if hi < lo:
                    raise RegexSyntaxError("reversed range")
                items.append((lo, hi))
            else:
                items += lo_set
        r = _norm(items)
        if self.icase:
            r = _case_fold(r, self.unicode)                   # folded first, negated second (regex-syntax's order)
        return _class_node(_negate(r, self.top) if neg else r, self.unicode)


def _reverse(n: Node) -> Node:
    if n.kind == "cat":
        return Node("cat", kids=[_reverse(k) for k in reversed(n.kids)])
    if n.kind in ("alt", "group"):
        return Node(n.kind, kids=[_reverse(k) for k in n.kids])
    if n.kind == "rep":
        return Node("rep", kids=[_reverse(n.kids[0])], lo=n.lo, hi=n.hi, greedy=n.greedy)
    return n   # lit / empty / assertions keep their meaning (start = start of haystack): they are decided on (left, right) contexts


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
    # state kinds: ("byte", set, nxt) ("split", [nxt...]) ("look", kind, nxt) ("match",)
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
        if k in _LOOKS:
            return self.add(("look", k, nxt))
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


# Look-around assertions and the context they look at.  A context is what lies on one side of a position: the edge of the
# haystack, a line feed, an ASCII word byte, or any other byte.
_LOOKS = ("start", "end", "sol", "eol", "wb", "nwb")      # \A ^ | \z $ | (?m)^ | (?m)$ | (?-u:\b) | (?-u:\B)
EDGE, LF, WORD, OTHER = 0, 1, 2, 3


def _ctx(byte: Optional[int]) -> int:
    if byte is None:
        return EDGE
    if byte == 0x0A:
        return LF
    if byte == 0x5F or 0x30 <= byte <= 0x39 or 0x41 <= byte <= 0x5A or 0x61 <= byte <= 0x7A:
        return WORD
    return OTHER


def _look_holds(kind: str, left: int, right: int) -> bool:
    if kind == "start":
        return left == EDGE
    if kind == "end":
        return right == EDGE
    if kind == "sol":
        return left in (EDGE, LF)
    if kind == "eol":
        return right in (EDGE, LF)
    same = (left == WORD) == (right == WORD)
    return not same if kind == "wb" else same


def _closure(nfa: _NFA, seeds: Sequence[int], sides: Optional[Tuple[int, int]] = None) -> Tuple[int, ...]:
    """Priority-ordered epsilon closure.  sides = (left, right): the contexts of the position are known and every assertion
    is decided (a failed one ends its thread); sides = None: the byte behind the position has not been seen yet — assertion
    states stay in the set and are decided by the next step, which is why match states are delayed by one byte."""
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
        elif t[0] == "look":
            if sides is None:
                out.append(s)
            elif _look_holds(t[1], sides[0], sides[1]):
                stack.append(t[2])
        else:
            out.append(s)
    return tuple(out)


def _can_match_empty(n: Node) -> bool:
    k = n.kind
    if k == "empty" or k in _LOOKS:
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
    kinds = {t[1] for t in nfa.st if t[0] == "look"}
    if kinds & {"sol", "eol"}:
        boundary[0x0A] = boundary[0x0B] = True
    if kinds & {"wb", "nwb"}:
        for b in (0x30, 0x3A, 0x41, 0x5B, 0x5F, 0x60, 0x61, 0x7B):
            boundary[b] = True
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

    # A DFA state: the NFA set (assertions still pending in it), "the previous position was a match" and — while an
    # assertion is pending — the context BEHIND the position in scan order (the byte scanned last, or the start configuration)
    kinds = {t[1] for t in nfa.st if t[0] == "look"}

    def behind(ctx: int, nfa_set: Tuple[int, ...]) -> int:
        if not any(nfa.st[x][0] == "look" for x in nfa_set):
            return OTHER
        if ctx == LF and not kinds & {"sol", "eol"}:
            ctx = OTHER
        if ctx == WORD and not kinds & {"wb", "nwb"}:
            ctx = OTHER
        return ctx

    Key = Tuple[Tuple[int, ...], bool, int]
    index: Dict[Key, int] = {}
    states: List[Key] = []
    table: List[List[int]] = []
    dead_key: Key = ((), False, OTHER)

    def intern(nfa_set: Tuple[int, ...], matched: bool, ctx: int) -> int:
        key = (nfa_set, matched, behind(ctx, nfa_set))
        if key == dead_key:
            return DEAD
        if key not in index:
            index[key] = len(states)
            states.append(key)
            table.append([DEAD] * alphabet_len)
        return index[key]

    states.append(dead_key); table.append([DEAD] * alphabet_len)   # state 0 = dead
    index[dead_key] = 0

    # Start::{NonWordByte, WordByte, Text, LineLF, LineCR, CustomLineTerminator}: what lies behind the search start
    START_CTX = (OTHER, WORD, EDGE, LF, OTHER, OTHER)
    starts = []
    for seed in ([un_start] if want_unanchored else [None]) + [start]:
        for cfg in range(6):
            if seed is None:
                starts.append(DEAD)
            else:
                starts.append(intern(_closure(nfa, [seed]), False, START_CTX[cfg]))

    done = 0
    while done < len(states):
        nfa_set, _flag, back = states[done]
        si = done
        done += 1
        if si == DEAD:
            continue

        def step(byte: Optional[int]) -> int:
            ahead = _ctx(byte)
            # the position between `back` and `ahead`: both sides known now (reverse scan: `back` is on the right)
            cur = _closure(nfa, list(nfa_set), (ahead, back) if reverse else (back, ahead))
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
            clo = _closure(nfa, nxt) if nxt else ()
            return intern(clo, matched, ahead)

        for c in range(ncls):
            table[si][c] = step(rep_byte[c])
        table[si][alphabet_len - 1] = step(None)
    return _Built(table, [f for (_s, f, _c) in states], starts, classes, alphabet_len)


def _serialize(b: _Built, *, start_kind: int, has_empty: bool, is_utf8: bool, always_anchored: bool) -> bytes:
    """dense::DFA::write_to, as the blobs regex-automata itself wrote lay it out (tests/golden/regex_automata_*.dfa): label,
    endianness, version, one unused u32, the flags as ONE u32 bit set, transition table, start table, match states, special,
    accelerators, quit set.  State order as there: dead, quit (never entered here: the quit set is empty), match states, the rest."""
    n = len(b.table)
    order = [0] + [i for i in range(1, n) if b.is_match[i]] + [i for i in range(1, n) if not b.is_match[i]]
    newidx = {old: (new if new == 0 else new + 1) for new, old in enumerate(order)}      # index 1 is the quit state
    stride2 = max(1, (b.alphabet_len - 1).bit_length())
    stride = 1 << stride2
    sid = lambda i: newidx[i] << stride2   # noqa: E731
    nm = sum(1 for i in range(1, n) if b.is_match[i])
    out = bytearray()
    out += LABEL + b"\0" * (32 - len(LABEL))
    out += struct.pack("<III", 0xFEFF, 2, 0)
    out += struct.pack("<I", int(has_empty) | (int(is_utf8) << 1) | (int(always_anchored) << 2))
    out += struct.pack("<II", n + 1, stride2)
    out += bytes(b.classes)
    tbl = [0] * ((n + 1) * stride)
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
    # special: max, quit_id, min_match, max_match, min_accel, max_accel, min_start, max_start
    quit_id = stride
    min_match = (2 << stride2) if nm else DEAD
    max_match = ((nm + 1) << stride2) if nm else DEAD
    out += struct.pack("<8I", max_match if nm else quit_id, quit_id, min_match, max_match, DEAD, DEAD, DEAD, DEAD)
    out += struct.pack("<I", 0)          # accelerators: none
    out += b"\0" * 32                    # quit set: empty
    return bytes(out)


def create_dfa(pattern: str, *, is_utf8: Optional[bool] = None, unicode: bool = False) -> DFA:
    """helpers/src/regex.rs:7-14 create_dfa: forward + reverse dense DFA blobs, padding stripped.
    unicode=True is what the reference builds (dfa::regex::Regex::new: Unicode classes, UTF-8 automata, flags.is_utf8 = 1);
    unicode=False is the byte-oriented mode ((?-u) throughout, flags.is_utf8 = 0 unless asked)."""
    if is_utf8 is None:
        is_utf8 = unicode
    ast = _Parser(pattern, unicode=unicode).parse()
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


def compile_regex_parts(parts: Sequence[RegexPattern], inp: bytes, *, unicode: bool = True) -> List[CompiledRegex]:
    """helpers/src/regex.rs:16-51: exactly one match of the pattern in the input, the capture groups named by
    capture_indices as (lossy UTF-8) strings, the DFA pair of create_dfa.  unicode=True is the reference's
    dfa::regex::Regex::new / meta::Regex::new (Unicode classes over UTF-8).  The match and the groups come from an
    independent engine: the `regex` module on the decoded input (its \\d \\w \\s are Unicode there), or Python's `re` on
    the bytes in byte mode."""
    out = []
    for part in parts:
        text = None
        if unicode:
            try:
                text = inp.decode("utf-8")
            except UnicodeDecodeError:
                if any(ord(ch) > 0x7F for ch in part.pattern) or _pyre.search(r"\\[dwsDWS]|\[\^|\.", part.pattern):
                    raise ValueError("the input is not UTF-8 and the pattern has Unicode-aware elements: no independent engine "
                                     "here can give its capture groups") from None
        if text is not None:
            try:
                import regex as _rx
                rx = _rx.compile(part.pattern)
            except ImportError:
                rx = _pyre.compile(part.pattern)
            ms = list(rx.finditer(text))
            grp = lambda m, gi: m.group(gi)   # noqa: E731
        else:
            rx = _pyre.compile(part.pattern.encode("utf-8"))
            ms = list(rx.finditer(inp))
            grp = lambda m, gi: None if m.group(gi) is None else m.group(gi).decode("utf-8", errors="replace")   # noqa: E731
        if len(ms) != 1:
            raise ValueError(f"Input doesn't match regex pattern: {part!r}")
        caps: List[str] = []
        for gi in part.capture_indices or []:
            if gi > rx.groups or ms[0].group(gi) is None:
                raise ValueError("Capture group not found")
            caps.append(grp(ms[0], gi))
        out.append(CompiledRegex(create_dfa(part.pattern, unicode=unicode), caps))
    return out
