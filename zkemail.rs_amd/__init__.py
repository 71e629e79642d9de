"""zkemail.rs_amd — MI355X-native batched e-mail verification witness engine.

Host-side mirror of ``zkemail_core``'s public surface (core/src/lib.rs:1-13) over the
C-ABI in ``include/zkemail_amd.h``.  All arithmetic on the verify path runs in
``libzkemail_amd.so`` (hand-written HIP for gfx950); this package only marshals buffers.
There is no CPU fallback: if the library or a GPU is missing, calls raise.
"""
from ._abi import (  # noqa: F401
    DFA, CompiledRegex, DebugBuffers, Email, EmailVerifierOutput, EmailWithRegex,
    EmailWithRegexVerifierOutput, ExternalInput, PackedBatch, PublicKey, RegexInfo,
    RESULT_DTYPE, STATUS_NAMES, STATUS_SITE,
)
from .engine import Engine, EngineError, VerifyPanic, verify_email, verify_email_with_regex  # noqa: F401

__all__ = [
    "Engine", "EngineError", "VerifyPanic", "verify_email", "verify_email_with_regex",
    "Email", "EmailWithRegex", "PublicKey", "ExternalInput", "DFA", "CompiledRegex", "RegexInfo",
    "EmailVerifierOutput", "EmailWithRegexVerifierOutput", "PackedBatch", "DebugBuffers",
]
