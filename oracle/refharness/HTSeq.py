"""Container-only stand-in for the third-party ``HTSeq`` package (TEST INFRASTRUCTURE).

The reference imports HTSeq unconditionally (SpliSER_v0_1_8.py:11) and uses exactly one
entry point, ``HTSeq.GFF_Reader(path)`` (SpliSER_v0_1_8.py:81-87), reading ``.type``,
``.name``, ``.iv.chrom``, ``.iv.start``, ``.iv.end`` and ``.iv.strand`` of each feature.
HTSeq is not vendored in /root/reference and no version is pinned there, so parity at this
boundary is *unpinned*: this stub follows HTSeq's documented GFF conventions
(0-based half-open intervals; feature name = value of the first attribute in column 9).

It exists only so the real reference can be executed in the build container to produce the
golden vectors under tests/golden/.  It is never imported by the product or on the GPU box.
"""


class _Interval(object):
    __slots__ = ("chrom", "start", "end", "strand")

    def __init__(self, chrom, start, end, strand):
        self.chrom = chrom
        self.start = start
        self.end = end
        self.strand = strand


class _Feature(object):
    __slots__ = ("type", "name", "iv")

    def __init__(self, ftype, name, iv):
        self.type = ftype
        self.name = name
        self.iv = iv


def _first_attribute_value(attr_field):
    first = attr_field.strip().split(";")[0].strip()
    if "=" in first:                      # GFF3  key=value
        value = first.split("=", 1)[1]
    elif " " in first:                    # GTF   key "value"
        value = first.split(" ", 1)[1]
    else:
        value = first
    return value.strip().strip('"')


class GFF_Reader(object):
    def __init__(self, path, end_included=True):
        self.path = path

    def __iter__(self):
        with open(self.path, "r") as handle:
            for raw in handle:
                if raw.startswith("#") or not raw.strip():
                    continue
                cols = raw.rstrip("\n").split("\t")
                if len(cols) < 9:
                    continue
                iv = _Interval(cols[0], int(cols[3]) - 1, int(cols[4]), cols[6])
                yield _Feature(cols[2], _first_attribute_value(cols[8]), iv)
