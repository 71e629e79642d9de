"""Cut serialised regex-automata dense DFAs (label "rust-regex-automata-dfa-dense", little-endian, version 2) out of a binary that
embeds them — Rust executables that use the bstr crate carry its precompiled whitespace automata verbatim.

    python tools/extract_regex_automata_blobs.py /path/to/executable tests/golden

The length of a blob is not stored anywhere: it is found by walking the layout (tests/test_regex_automata_blobs.py `parse`);
a layout mistake would not end exactly where the next object starts.  This is how tests/golden/regex_automata_ws_anchored_*.dfa
were made (sha256 661745fc…daaac7 / 79182531…8e5f58); the executable itself is not part of this repository and is not needed again."""
import hashlib
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_regex_automata_blobs import parse        # noqa: E402

data = open(sys.argv[1], "rb").read()
for k, m in enumerate(re.finditer(rb"rust-regex-automata-dfa-dense\0\0\0\xff\xfe\0\0\x02\0\0\0", data)):
    d, used = parse(data[m.start():m.start() + (1 << 24)])
    blob = data[m.start():m.start() + used]
    kind = {0: "both", 1: "unanchored", 2: "anchored"}[d["start_kind"]]
    out = os.path.join(sys.argv[2], f"regex_automata_{k}_{kind}_{d['state_len']}states.littleendian.dfa")
    open(out, "wb").write(blob)
    print(out, used, hashlib.sha256(blob).hexdigest())
