"""Host-side input partitioning: the contract of the reference's partitioned_file<...,FASTQParser>
(src/io/file.hpp:1216-1430) for one node -- split a FASTQ buffer into `n` byte ranges that each
start at a true record start, using the 4-line rule of FASTQParser::find_first_record
(src/io/fastq_loader.hpp:269-364): after the first EOL at or after the nominal split point look at
the first characters of the next four lines; a record starts at the line where '@' is followed two
lines later by '+'."""
import numpy as np

_EOL = (10, 13)


def _is_eol(b):
    return b == 10 or b == 13


def find_first_record(data, pos):
    """first record start at or after byte `pos` (len(data) if none)"""
    n = len(data)
    if pos <= 0:
        return 0
    i = pos
    # not at an EOL: skip the rest of this (partial) line
    while i < n and not _is_eol(data[i]):
        i += 1
    starts, firsts = [], []
    for _ in range(4):
        while i < n and _is_eol(data[i]):
            i += 1
        if i >= n:
            return n
        starts.append(i)
        firsts.append(data[i])
        while i < n and not _is_eol(data[i]):
            i += 1
    at, plus = ord("@"), ord("+")
    if firsts[0] == at and firsts[2] == plus:
        return starts[0]
    if firsts[1] == at and firsts[3] == plus:
        return starts[1]
    if firsts[0] == plus and firsts[2] == at:
        return starts[2]
    if firsts[1] == plus and firsts[3] == at:
        return starts[3]
    return n


def partition_fastq(data, n_parts):
    """record-aligned [begin, end) byte ranges, one per rank; they tile the buffer exactly"""
    data = memoryview(data) if not isinstance(data, np.ndarray) else data
    n = len(data)
    cuts = [find_first_record(data, (n * r) // n_parts) for r in range(n_parts)] + [n]
    for r in range(1, n_parts + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(n_parts)]
