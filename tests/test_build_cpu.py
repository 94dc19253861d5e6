"""selfreconcode_amd.build: the library carries the digest of the sources it was built from, a library built from other sources is
never loaded silently, and without a compiler the failure says what is wrong (there is no CPU fallback to fall back to)."""
import os

import pytest

from selfreconcode_amd import build


def test_library_in_tree_was_built_from_these_sources():
    assert build.build_lib(verbose=False) == build.LIB                       # (builds on a fresh checkout, as tests/test_abi.py does; a no-op otherwise)
    assert os.path.isfile(build.LIB)
    assert build.embedded_digest() == build._digest() and not build.is_stale()


def test_embedded_digest_reads_the_marker(tmp_path):
    assert build.embedded_digest(str(tmp_path / "missing.so")) is None
    plain = tmp_path / "plain.so"
    plain.write_bytes(b"\x7fELF" + b"\0" * 100)
    assert build.embedded_digest(str(plain)) is None                          # a library that predates the marker
    marked = tmp_path / "marked.so"
    marked.write_bytes(b"\x7fELF" + b"\0" * 10 + build.MARKER + b"ab" * 32 + b"\0tail")
    assert build.embedded_digest(str(marked)) == "ab" * 32


def test_digest_follows_sources_and_flags(monkeypatch):
    d0 = build._digest()
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DX=1"])
    assert build._digest() != d0                                              # another flag set is another library


def test_stale_library_without_a_compiler_fails_loudly(monkeypatch):
    monkeypatch.setattr(build, "_digest", lambda: "0" * 64)                   # pretend the sources changed
    monkeypatch.setattr(build, "HIPCC", "/nonexistent/hipcc")
    assert build.is_stale()
    with pytest.raises(RuntimeError, match="built from other sources"):
        build.build_lib(verbose=False)
