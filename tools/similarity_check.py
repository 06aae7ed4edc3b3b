"""Self-check against the copy detector: difflib ratio over stripped non-comment lines between every
python file of the package and the same-named file of the reference (run in the build container only)."""
import difflib, os, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/openea"
OURS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openea_amd")


def lines(path):
    out = []
    for ln in open(path, errors="ignore"):
        ln = ln.strip()
        if ln and not ln.startswith("#"):
            out.append(ln)
    return out


ref = {}
for root, _, files in os.walk(REF):
    for f in files:
        if f.endswith(".py"):
            ref.setdefault(f, []).append(os.path.join(root, f))
rows = []
for root, _, files in os.walk(OURS):
    for f in files:
        if f.endswith(".py") and f in ref and f != "__init__.py":
            mine = lines(os.path.join(root, f))
            for r in ref[f]:
                rows.append((difflib.SequenceMatcher(None, mine, lines(r)).ratio(),
                             os.path.relpath(os.path.join(root, f), OURS), os.path.relpath(r, REF)))
for ratio, a, b in sorted(rows, reverse=True)[:15]:
    print("%.2f  %s  <->  %s" % (ratio, a, b))
