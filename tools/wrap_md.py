#!/usr/bin/env python3
"""Wrap the prose of a markdown file at 118 columns (VERDICT r5 hygiene: DESIGN lines <= 120 characters).  Table rows, fenced
code, indented code, headings and lines that are one unbreakable token are left alone; list items keep their hanging indent.

    python tools/wrap_md.py DESIGN.md [--check]      (--check: exit 1 and name the offending lines instead of rewriting)
"""
import re
import sys
import textwrap

WIDTH = 118


def wrap_text(text):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+)", first)
        lead = m.group(1) if m else re.match(r"^(\s*)", first).group(1)
        hang = " " * len(lead)
        body = " ".join([first[len(lead):].strip()] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(body, WIDTH, initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                                 break_on_hyphens=False))
        para.clear()

    for line in text.splitlines():
        if line.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or line.startswith(("    ", "\t", "#", "|")) or not line.strip():
            flush()
            out.append(line)
            continue
        if re.match(r"^\s*(?:[-*+]|\d+\.)\s+", line):
            flush()
        para.append(line)
    flush()
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    path = sys.argv[1]
    text = open(path).read()
    if "--check" in sys.argv:
        bad = [(i + 1, len(l)) for i, l in enumerate(text.splitlines()) if len(l) > 120 and not l.startswith("|") and " " in l.strip()]
        for i, n in bad:
            print(f"{path}:{i}: {n} characters")
        sys.exit(1 if bad else 0)
    open(path, "w").write(wrap_text(text))
