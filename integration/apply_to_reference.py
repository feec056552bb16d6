#!/usr/bin/env python
"""The whole reference-side change of INTEGRATION.md section 2, applied to a checkout of MenghanXia/DisentangledColorization:

    python integration/apply_to_reference.py /path/to/DisentangledColorization [--revert]

    main/colorizer/inference.py:17   `import model, basic`  ->  `import basic` + `from disentangledcolorization_amd import model`
    main/spixelseg/inference.py:16   `import model, basic`  ->  `from disentangledcolorization_amd import model, basic`

Nothing else in the reference changes (the nn.DataParallel wrapping of inference.py:76-82 stays: replicas share the native context).  The
files keep their line endings (the checkout has CRLF); running it twice is harmless.  `disentangledcolorization_amd` must be importable
(PYTHONPATH = this repository's root) and built (python -m disentangledcolorization_amd.build)."""
import os
import sys

EDITS = {
    os.path.join("main", "colorizer", "inference.py"): ("import model, basic", "import basic\nfrom disentangledcolorization_amd import model"),
    os.path.join("main", "spixelseg", "inference.py"): ("import model, basic", "from disentangledcolorization_amd import model, basic"),
}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 1:
        raise SystemExit(__doc__)
    revert = "--revert" in sys.argv
    for rel, (old, new) in EDITS.items():
        path = os.path.join(args[0], rel)
        with open(path, "rb") as f:
            raw = f.read()
        eol = b"\r\n" if b"\r\n" in raw else b"\n"
        lines = raw.split(eol)
        old_l, new_l = [old.encode()], [s.encode() for s in new.split("\n")]
        src, dst = (new_l, old_l) if revert else (old_l, new_l)
        for i in range(len(lines) - len(src) + 1):
            if lines[i:i + len(src)] == src:
                lines[i:i + len(src)] = dst
                with open(path, "wb") as f:
                    f.write(eol.join(lines))
                print("%s:%d  %s" % (rel, i + 1, "reverted" if revert else "patched"))
                break
        else:
            state = dst
            print("%s: %s" % (rel, "already done" if any(lines[i:i + len(state)] == state for i in range(len(lines))) else "import line not found"))


if __name__ == "__main__":
    main()
