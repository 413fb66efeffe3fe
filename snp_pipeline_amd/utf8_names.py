"""Pileups whose contig names are not plain ASCII.

The reference reads a pileup as text (pileup.py:405: the locale's encoding, UTF-8 on the pipeline's platforms), splits every line
on white space and compares the first field with the names of the snplist — so a contig called ``chrä`` or ``染色体1`` is just a
name to it.  The device kernels work on bytes and keep their SWAR tests cheap by assuming ASCII (a byte >= 0x80 anywhere is
reported as scan code 3).  This module bridges the two for the case that matters: a file that is valid UTF-8 and has its
non-ASCII characters in contig names only.  Every name — in the pileup's first column and in the site lists alike — goes through
an ESCAPE that is injective, keeps the bytewise order of names, and yields plain ASCII: a byte b >= 0x7E becomes ``~`` and two
characters of ``0123456789:;<=>?`` spelling b - 0x7E in base 16.  The device then sees an ASCII pileup whose names compare and sort
exactly as the originals do; the CHROM column of consensus.vcf is spelled back afterwards.  No arithmetic happens here.

Still refused (what the characters would MEAN differs between text and bytes there): non-ASCII characters in any other column
(a multi-byte read base or quality is one symbol to the reference and several bytes to the kernels), and Unicode white space
anywhere (str.split() breaks a field at U+00A0, U+2028 ...; the kernels split at ASCII white space).
"""
import os
import re
import tempfile

import numpy as np

_WS = np.array([9, 10, 11, 12, 13, 32, 28, 29, 30, 31], dtype=np.uint8)       # what str.split() removes from ASCII text
_UNICODE_SPACE = re.compile("[\x85\xa0  -     　]")
CHUNK = 16 << 20           # bytes per read block (its index arrays take ~16 x that for a moment)


class Refused(ValueError):
    """The file is valid UTF-8 but uses non-ASCII characters where text and bytes part ways."""


def escape_name(name):
    """bytes -> ASCII bytes; order-preserving and injective (see the module docstring)."""
    if not any(b >= 0x7E for b in name):
        return name
    out = bytearray()
    for b in name:
        if b >= 0x7E:
            out += bytes((0x7E, 0x30 + ((b - 0x7E) >> 4), 0x30 + ((b - 0x7E) & 15)))
        else:
            out.append(b)
    return bytes(out)


def unescape_name(name):
    if b"~" not in name:
        return name
    out, i = bytearray(), 0
    while i < len(name):
        if name[i] == 0x7E and i + 2 < len(name):                # "~" and its two digits
            out.append(0x7E + ((name[i + 1] - 0x30) << 4) + (name[i + 2] - 0x30))
            i += 3
        else:
            out.append(name[i])
            i += 1
    return bytes(out)


def escape_names(names):
    """[str] -> [str]: the names of a site list as the device will see them."""
    return [escape_name(n.encode("utf-8")).decode("ascii") for n in names]


def _escape_chunk(raw):
    """One stretch of whole lines.  Returns the escaped bytes (or `raw` itself when nothing needs escaping)."""
    text = raw.decode("utf-8")                                   # UnicodeDecodeError where the reference's text-mode read raises it
    arr = np.frombuffer(raw, dtype=np.uint8)
    hi = arr >= 0x80
    if not hi.any() and not (arr == 0x7E).any():
        return raw
    if hi.any() and _UNICODE_SPACE.search(text):
        raise Refused("Unicode white space in the pileup: str.split() and the device's ASCII split would disagree")
    ws = np.isin(arr, _WS)
    term = (arr == 10) | (arr == 13)
    starts = np.concatenate(([0], np.flatnonzero(term) + 1))
    nonws_pos, ws_pos = np.flatnonzero(~ws), np.flatnonzero(ws)
    k = np.searchsorted(nonws_pos, starts)
    f0b = np.unique(nonws_pos[k[k < len(nonws_pos)]])            # where a first field begins
    j = np.searchsorted(ws_pos, f0b)
    f0e = np.full(len(f0b), len(arr), dtype=np.int64)            # ... and ends: at the next white space, or with the data
    if len(ws_pos):
        inside = j < len(ws_pos)
        f0e[inside] = ws_pos[j[inside]]
    delta = np.zeros(len(arr) + 1, dtype=np.int32)
    np.add.at(delta, f0b, 1)
    np.add.at(delta, f0e, -1)
    in_f0 = np.cumsum(delta[:-1]) > 0
    if (hi & ~in_f0).any():
        raise Refused("non-ASCII characters outside the contig-name column")
    need = in_f0 & (arr >= 0x7E)
    if not need.any():
        return raw
    rep = np.where(need, 3, 1).astype(np.int64)
    out = np.repeat(arr, rep)
    at = (np.cumsum(rep) - rep)[need]
    v = arr[need].astype(np.int32) - 0x7E
    out[at] = 0x7E
    out[at + 1] = (0x30 + (v >> 4)).astype(np.uint8)
    out[at + 2] = (0x30 + (v & 15)).astype(np.uint8)
    return out.tobytes()


def _copy_directory(pileup_path):
    """Where the escaped copy goes: $TMPDIR when the user names one, else beside the pileup (the copy is as large as the file —
    many GB — and the default /tmp is often a small tmpfs), else wherever tempfile puts things."""
    if os.environ.get("TMPDIR"):
        return None                                              # tempfile honours it
    beside = os.path.dirname(os.path.abspath(pileup_path))
    return beside if os.access(beside, os.W_OK | os.X_OK) else None


def escaped_copy(pileup_path, directory=None):
    """Write the pileup with escaped contig names to a temporary file and return its path.  Raises UnicodeDecodeError for a
    file that is not valid UTF-8 (as the reference's read does), Refused for one this bridge cannot carry, OSError (ENOSPC ...)
    when the copy cannot be written."""
    fd, tmp = tempfile.mkstemp(prefix=".snpgpu_names_", suffix=".pileup", dir=directory or _copy_directory(pileup_path))
    ok = False
    try:
        with os.fdopen(fd, "wb") as out, open(pileup_path, "rb") as f:
            carry = b""
            while True:
                block = f.read(CHUNK)
                if not block:
                    if carry:
                        out.write(_escape_chunk(carry))
                    break
                block = carry + block
                cut = max(block.rfind(b"\n"), block.rfind(b"\r")) + 1      # whole lines only (a character never straddles a line end)
                if cut == 0:
                    carry = block
                    continue
                if block[cut - 1:cut] == b"\r":                            # the "\n" of a CR LF pair may open the next block: keep the pair together
                    cut -= 1
                    if cut == 0:
                        carry = block
                        continue
                out.write(_escape_chunk(block[:cut]))
                carry = block[cut:]
        ok = True
        return tmp
    finally:
        if not ok:
            os.unlink(tmp)


def unescape_vcf_chrom(vcf_path):
    """Spell the CHROM column of the data lines back (the file was written from escaped names): line by line into a temporary
    file beside it, which then takes its place in one step — with --vcfAllPos the file has a row per pileup line, and a crash in the
    middle must not leave half a consensus.vcf that is newer than its inputs."""
    tmp = "%s.names.%d" % (vcf_path, os.getpid())
    changed = False
    try:
        with open(vcf_path, "rb") as f, open(tmp, "wb") as out:
            for ln in f:
                if ln[:1] != b"#" and b"\t" in ln:
                    chrom, rest = ln.split(b"\t", 1)
                    if b"~" in chrom:
                        ln = unescape_name(chrom) + b"\t" + rest
                        changed = True
                out.write(ln)
        if changed:
            os.replace(tmp, vcf_path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
