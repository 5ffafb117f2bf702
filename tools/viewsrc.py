#!/usr/bin/env python3
"""Reading aid: print a C++ source range with line numbers, hiding the bodies of
debug-only preprocessor blocks (#ifdef *_EN_DBG_OUT ... #endif) and blank lines.
Usage: viewsrc.py FILE [START [END]]"""
import sys, re
fn = sys.argv[1]
start = int(sys.argv[2]) if len(sys.argv) > 2 else 1
end = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
stack = []  # entries: True if hidden
out = []
for i, line in enumerate(open(fn, encoding='utf-8-sig', errors='replace'), 1):
    s = line.strip()
    m = re.match(r'#\s*(ifdef|ifndef|if)\b(.*)', s)
    if m:
        hide = m.group(1) == 'ifdef' and re.search(r'(EN_DBG_OUT|_VERBOSE|LB_EN_PIXEL_DBG)', m.group(2)) is not None
        stack.append(hide)
        if hide or any(stack[:-1]):
            continue
    elif re.match(r'#\s*else', s) and stack:
        if stack[-1] is True and not any(stack[:-1]):
            stack[-1] = False
            continue
        elif stack[-1] is False and False:
            pass
    elif re.match(r'#\s*endif', s) and stack:
        h = stack.pop()
        if h or any(stack):
            continue
        # if it was a flipped (else-shown) block we also skip the endif silently
        pass
    if any(stack):
        continue
    if not s:
        continue
    if start <= i <= end:
        out.append(f"{i}:{line.rstrip()}")
print("\n".join(out))
