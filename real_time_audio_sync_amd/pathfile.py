"""The reference's alignment-path text files (livenote_live.py:138-154, wtw_live.py:169-174,208-210;
reader tests.py:20-27): five CRLF header lines -- reference WAV path, ``fft_len: N``, ``hop_size: N``
and two algorithm parameters -- followed by ``"<live_frame> <ref_frame>"`` lines.  Files written by
test_simple.py:183-185 (e.g. Songs/chopin/tests/wtw_test_20b.txt) have no header."""


def write_path_file(filename, path, ref=None, fft_len=4096, hop_size=2048, params=None):
    """``params``: ordered pairs for the two parameter lines, e.g. [('search_band_width', 50),
    ('max_run_count', 3)] or [('dtw_win_size', 40960), ('dtw_hop_size', 20480)].  ``ref=None`` writes
    the header-less form."""
    with open(filename, "w", newline="") as fh:
        if ref is not None:
            fh.write("%s\r\n" % ref)
            fh.write("fft_len: %d\r\n" % fft_len)
            fh.write("hop_size: %d\r\n" % hop_size)
            for name, value in (params or [("search_band_width", 0), ("max_run_count", 0)]):
                fh.write("%s: %d\r\n" % (name, value))
        for l, r in path:
            fh.write("%d %d\r\n" % (l, r))


def read_path_file(filename, header_lines=5):
    """-> list of (live_frame, ref_frame).  header_lines=5 is tests.py:data_from_file; use 0 for
    header-less files.  Trailing non-numeric lines (wtw_live.py's "Percent incorrect ...") are skipped."""
    path = []
    with open(filename, newline="") as fh:
        lines = fh.read().splitlines()
    for line in lines[header_lines:]:
        tok = line.strip().split("\t")[0].split(" ")
        if len(tok) >= 2 and tok[0].lstrip("-").isdigit() and tok[1].lstrip("-").isdigit():
            path.append((int(tok[0]), int(tok[1])))
    return path


def read_header(filename):
    """-> dict(ref=..., fft_len=..., hop_size=..., params=[(name, value), (name, value)])."""
    with open(filename, newline="") as fh:
        lines = fh.read().splitlines()[:5]
    kv = [ln.split(":") for ln in lines[1:]]
    return dict(ref=lines[0].strip(), fft_len=int(kv[0][1]), hop_size=int(kv[1][1]),
                params=[(k.strip(), int(v)) for k, v in kv[2:]])
