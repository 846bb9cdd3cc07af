#!/usr/bin/env python3
"""docs/reflow_md.py FILE... -- re-wrap paragraphs and bullets of a markdown file at 118 columns (tables, headings, code fences and
blank lines stay as they are).  DESIGN.md is kept below 120 columns with it."""
import sys
import textwrap


def reflow(path):
    lines = open(path, encoding="utf8").read().split("\n")
    out, buf, bullet, fence = [], [], False, False

    def flush():
        nonlocal buf, bullet
        if buf:
            text = " ".join(x.strip() for x in buf)
            out.extend(textwrap.wrap(text, 118, subsequent_indent="  " if bullet else "", break_long_words=False, break_on_hyphens=False))
        buf, bullet = [], False

    for ln in lines:
        if ln.startswith("```"):
            flush()
            fence = not fence
            out.append(ln)
        elif fence or ln.startswith("|") or ln.startswith("#") or ln.startswith("<!--") or ln.strip() == "":
            flush()
            out.append(ln)
        elif ln.startswith("* "):
            flush()
            bullet, buf = True, [ln]
        else:
            buf.append(ln)
    flush()
    open(path, "w", encoding="utf8").write("\n".join(out))
    return sum(1 for ln in out if len(ln) > 120 and not ln.startswith("|"))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p, "lines over 120 columns outside tables:", reflow(p))
