"""Re-join and re-wrap the prose of a markdown file at word boundaries (default 118 columns): paragraphs and list items become whole
sentences again; headings, tables, fenced code and indented code are left alone.   python scripts/reflow_md.py DESIGN.md [width]"""
import re, sys, textwrap

def reflow(text, width=118):
    out, block, fence = [], [], False
    def flush():
        if not block:
            return
        items, cur = [], None
        for ln in block:
            m = re.match(r"^(\s*)([*\-]|\d+\.)\s+", ln)
            if m:
                cur = [m.group(0), ln[m.end():].strip()]
                items.append(cur)
            elif cur is None:
                cur = ["", ln.strip()]
                items.append(cur)
            else:
                cur[1] += " " + ln.strip()
        for first, body in items:
            indent = " " * len(first)
            body = re.sub(r"(?<=[^ ]) (?=[^ ])", " ", body)
            out.extend(textwrap.wrap(body, width=width, initial_indent=first, subsequent_indent=indent, break_long_words=False,
                                     break_on_hyphens=False) or [first.rstrip()])
        block.clear()
    for ln in text.split("\n"):
        if ln.startswith("```"):
            flush(); fence = not fence; out.append(ln); continue
        if fence or ln.startswith("|") or ln.startswith("#") or ln.startswith("    ") and not block or not ln.strip():
            flush(); out.append(ln); continue
        block.append(ln)
    flush()
    return "\n".join(out)

if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
