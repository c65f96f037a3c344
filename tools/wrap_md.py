#!/usr/bin/env python3
"""usage: wrap_md.py <file.md> [width]  -- re-wraps paragraphs and list items of a markdown file to <= width characters
(default 118); headings, tables, code blocks (fenced or indented by four spaces) and blank lines are left as they are.
Reports the lines that are still longer (tables)."""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path, encoding="utf-8").read().split("\n")
out, para, indent_first, indent_rest = [], [], "", ""
in_fence = False


def flush():
    global para
    if para:
        text = " ".join(s.strip() for s in para)
        out.extend(textwrap.wrap(text, width=width, initial_indent=indent_first, subsequent_indent=indent_rest,
                                 break_long_words=False, break_on_hyphens=False))
        para = []


for ln in lines:
    if ln.startswith("```"):
        flush()
        in_fence = not in_fence
        out.append(ln)
        continue
    if in_fence or ln.startswith("    ") and not para or ln.startswith("|") or ln.startswith("#") or not ln.strip():
        flush()
        out.append(ln)
        continue
    m = re.match(r"^(\s*)([*\-] |\d+\. )(.*)$", ln)
    if m:
        flush()
        indent_first = m.group(1) + m.group(2)
        indent_rest = m.group(1) + " " * len(m.group(2))
        para = [m.group(3)]
        continue
    if not para:
        indent_first = indent_rest = re.match(r"^(\s*)", ln).group(1)
    para.append(ln)
flush()
open(path, "w", encoding="utf-8").write("\n".join(out))
long = [(i + 1, len(l)) for i, l in enumerate(out) if len(l) > 120]
for i, n in long:
    print(f"{path}:{i}: {n} characters")
