"""Resolve `#ifdef X / #ifndef X / #else / #endif` blocks for macros whose value is now fixed (round 6's pruning of closed
experiment knobs): python tools/resolve_ifdefs.py <file> NAME=defined|undefined ...   Only simple ifdef/ifndef forms."""
import re
import sys

path = sys.argv[1]
state = dict(a.split("=") for a in sys.argv[2:])
out, stack = [], []  # stack entries: (keep_this_branch or None when the block is not ours, parent_keep)
keep = True
for line in open(path).read().split("\n"):
    m = re.match(r"\s*#\s*(ifdef|ifndef)\s+(\w+)", line)
    if m and m.group(2) in state:
        defined = state[m.group(2)] == "defined"
        take = defined if m.group(1) == "ifdef" else not defined
        stack.append((take, keep))
        keep = keep and take
        continue
    if re.match(r"\s*#\s*if", line):
        stack.append((None, keep))
        if keep:
            out.append(line)
        continue
    if re.match(r"\s*#\s*else", line) and stack:
        take, parent = stack[-1]
        if take is None:
            if keep:
                out.append(line)
        else:
            stack[-1] = (not take, parent)
            keep = parent and (not take)
        continue
    if re.match(r"\s*#\s*endif", line) and stack:
        take, parent = stack.pop()
        if take is None:
            if keep:
                out.append(line)
        keep = parent
        continue
    if keep:
        out.append(line)
open(path, "w").write("\n".join(out))
