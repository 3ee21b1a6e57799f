"""One-off (round 5): DESIGN.md (142 KB, median line several hundred characters) -> docs/design/*.md with lines of <= 120 characters.
No content change: paragraphs and list items are re-wrapped; a table whose rows do not fit 120 characters becomes a list of its rows
(first cell = the item's lead, the other cells follow as "*column:* text").  python tools/split_design.py OLD_DESIGN.md"""
import os
import re
import sys

W = 120  # (bytes, not characters: the notes are full of multi-byte symbols)
src = open(sys.argv[1]).read().split("\n")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blen(s):
    return len(s.encode())


def wrap(text, first="", rest=""):
    """greedy wrap to W BYTES (words are never broken)"""
    lines, cur, empty = [], first, True
    for w in text.split():
        cand = cur + ("" if empty else " ") + w
        if empty or blen(cand) <= W:
            cur, empty = cand, False
        else:
            lines.append(cur)
            cur = rest + w
    lines.append(cur if not empty else first.rstrip())
    return lines


def cells(row):
    return [c.strip() for c in row.strip().strip("|").split("|")]


def table(rows):
    if all(blen(r) <= W for r in rows):
        return rows
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        cs = cells(r)
        out += wrap(cs[0] if cs[0].startswith("**") else f"**{cs[0]}**", "* ", "  ")
        for h, c in zip(head[1:], cs[1:]):
            if c and c != "—":
                out += wrap(f"*{h}:* {c}", "  ", "  ")
        out.append("")
    return out


def convert(lines):
    out, i = [], 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("```"):
            j = i + 1
            while j < len(lines) and not lines[j].startswith("```"):
                j += 1
            out += lines[i:j + 1]
            i = j + 1
        elif l.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            out += table(lines[i:j])
            i = j
        elif l.startswith("#") or not l.strip():
            out.append(l)
            i += 1
        else:
            m = re.match(r"^(\s*(?:[*\-]|\d+\.)\s+)", l)
            first = m.group(1) if m else re.match(r"^\s*", l).group(0)
            rest = " " * len(first)
            para = [l[len(first):] if m else l.strip()]
            j = i + 1
            while j < len(lines) and lines[j].strip() and not re.match(r"^(\s*(?:[*\-]|\d+\.)\s+|#|\||```|\s+\*[^*]+:\*\s)", lines[j]):
                para.append(lines[j].strip())
                j += 1
            out += wrap(" ".join(para), first, rest)
            i = j
    return out


def section(start_pat, end_pat=None):
    a = next(i for i, l in enumerate(src) if re.match(start_pat, l))
    b = next((i for i, l in enumerate(src) if i > a and end_pat and re.match(end_pat, l)), len(src))
    return src[a:b]


FILES = {
    "boundary.md": [(r"^## 1\. ", r"^## 2\. ")],
    "layout.md": [(r"^## 2\. ", r"^## 3\. ")],
    "kernels.md": [(r"^## 3\. ", r"^### K2′ with a search width")],
    "k2_graph.md": [(r"^### K2′ with a search width", r"^### K4b")],
    "k4_bruteforce.md": [(r"^### K4b", r"^### C: the exactness certificate")],
    "certificate.md": [(r"^### C: the exactness certificate", r"^### Tuning switches")],
    "switches.md": [(r"^### Tuning switches", r"^## 4\. ")],
    "oracle.md": [(r"^## 4\. ", r"^## 5\. ")],
    "measurement.md": [(r"^## 5\. ", r"^## 6\. ")],
    "multi_gpu.md": [(r"^## 6\. ", r"^## 7\. ")],
    "scope_and_next.md": [(r"^## 7\. ", r"^## 9\. ")],
    "history.md": [(r"^## 9\. ", None)],
}
for name, parts in FILES.items():
    body = []
    for a, b in parts:
        body += section(a, b)
    text = "\n".join(convert(body)).rstrip() + "\n"
    text = f"<!-- part of the design notes: index in /DESIGN.md (split from the single file of rounds 1-4; content unchanged) -->\n\n" + text
    open(os.path.join(root, "docs", "design", name), "w").write(text)
    long = sum(1 for l in text.split("\n") if blen(l) > W)
    print(f"{name}: {len(text.split(chr(10)))} lines, {long} over {W}")
