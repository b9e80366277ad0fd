"""One-off restructuring of DESIGN.md: the long sections move to docs/*.md files of <= 200 lines with lines of
<= 160 columns; DESIGN.md keeps the scope table (one line per row + link), parity, boundary, layout, protocol, a short kernel table, the
host / device flow, multi-GPU and an index.  Tables whose rows do not fit 160 columns become lists (first cell bold, the other cells
labelled by the table's header).  Kept in the tree so that the transformation can be re-read; it is not part of the product.
Usage: python3 tools/split_design.py DESIGN.md   (rewrites DESIGN.md and docs/*.md in place; run once on the round-4 file)"""
import os
import re
import sys
import textwrap

W = 160
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def wrap(text, first="", rest=""):
    return textwrap.fill(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells_of(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    out, cur, tick = [], "", False
    for ch in row:
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def reflow(lines):
    """markdown lines -> markdown lines, none longer than W (except inside code fences and unbreakable tokens)"""
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i]
        if ln.startswith("```"):
            out.append(ln)
            i += 1
            while i < n and not lines[i].startswith("```"):
                out.append(lines[i])
                i += 1
            if i < n:
                out.append(lines[i])
                i += 1
            continue
        if ln.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|[\s:|-]+\|\s*$", lines[i + 1]):
            j = i
            while j < n and lines[j].lstrip().startswith("|"):
                j += 1
            table = lines[i:j]
            if max(len(x) for x in table) <= W:
                out += table
            else:
                head = cells_of(table[0])
                for row in table[2:]:
                    c = cells_of(row)
                    out.append(wrap("**" + c[0].strip("* ") + "**", "* ", "  "))
                    for k in range(1, len(c)):
                        if not c[k] or c[k] == "-":
                            continue
                        label = head[k] if k < len(head) and len(c) == len(head) else ""
                        out.append(wrap((label + ": " if label else "") + c[k], "  - ", "    "))
            i = j
            continue
        if len(ln) <= W or ln.startswith("#"):
            out.append(ln)
            i += 1
            continue
        m = re.match(r"^(\s*)([*-]|\d+\.)\s+", ln)
        if m:
            ind = m.group(1)
            out.append(wrap(ln[m.end():], ind + m.group(2) + " ", ind + " " * (len(m.group(2)) + 1)))
        else:
            ind = re.match(r"^\s*", ln).group(0)
            out.append(wrap(ln.strip(), ind, ind))
        i += 1
    return "\n".join(out).split("\n")


def split_parts(lines, limit=195):
    """[lines] -> list of parts of <= limit lines, cut at blank lines (preferring a header right after)"""
    parts = []
    while len(lines) > limit:
        cut = None
        for k in range(limit, limit // 2, -1):
            if lines[k].startswith("#") and lines[k - 1].strip() == "":
                cut = k
                break
        if cut is None:
            for k in range(limit, limit // 2, -1):
                if lines[k].strip() == "" and not lines[k + 1].startswith((" ", "\t")):
                    cut = k + 1
                    break
        if cut is None:
            for k in range(limit, limit // 2, -1):
                if lines[k].strip() == "":
                    cut = k + 1
                    break
        if cut is None:
            cut = limit
        parts.append(lines[:cut])
        lines = lines[cut:]
    parts.append(lines)
    return parts


def write_doc(name, title, intro, body_lines):
    body = reflow(body_lines)
    parts = split_parts(body)
    names = [name if k == 0 else "%s_%d" % (name, k + 1) for k in range(len(parts))]
    for k, part in enumerate(parts):
        head = ["# " + title + (" (part %d of %d)" % (k + 1, len(parts)) if len(parts) > 1 else ""), ""]
        if k == 0 and intro:
            head += reflow([intro]) + [""]
        if len(parts) > 1:
            nav = []
            if k > 0:
                nav.append("previous: [%s.md](%s.md)" % (names[k - 1], names[k - 1]))
            if k + 1 < len(parts):
                nav.append("next: [%s.md](%s.md)" % (names[k + 1], names[k + 1]))
            head += ["(" + "; ".join(nav) + "; back to [DESIGN.md](../DESIGN.md))", ""]
        else:
            head += ["(back to [DESIGN.md](../DESIGN.md))", ""]
        with open(os.path.join(ROOT, "docs", names[k] + ".md"), "w") as f:
            f.write("\n".join(head + part).rstrip("\n") + "\n")
    return names


def main(path):
    text = open(path).read().split("\n")
    # sections by "## " headers
    idx = [i for i, l in enumerate(text) if l.startswith("## ")]
    sec = {}
    for a, b in zip(idx, idx[1:] + [len(text)]):
        m = re.match(r"## (\d+)\.", text[a])
        sec[int(m.group(1)) if m else text[a]] = (text[a], text[a + 1:b])
    pre = text[:idx[0]]
    os.makedirs(os.path.join(ROOT, "docs"), exist_ok=True)
    moved = {}

    def move(num, name, title):
        h, body = sec[num]
        moved[num] = (write_doc(name, title, "Moved from DESIGN.md section %d (round 5).  Citations `path:line` are relative to `/root/reference/` unless they start with a repo directory." % num, body), h)

    # section 0: the table rows become docs/scope.md; DESIGN keeps one line per row
    h0, body0 = sec[0]
    rows = [l for l in body0 if l.startswith("|")][2:]
    tail0 = [l for l in body0 if not l.startswith("|")]
    scope_lines, short = [], []
    for k, row in enumerate(rows):
        c = cells_of(row)
        what = c[0]
        state = c[-1]
        where = c[-2] if len(c) >= 3 else ""
        mid = c[1] if len(c) == 4 else ""
        anchor = "row-%d" % (k + 1)
        scope_lines += ["## Row %d" % (k + 1), "", "**" + what + "**", ""]
        if mid:
            scope_lines += ["* What: " + mid]
        scope_lines += ["* Where it lives here: " + where, "* State: " + state, ""]
        plain = state.replace("**", "")
        first = re.split(r"(?<=[a-z0-9)`*])[;:.] ", plain, maxsplit=1)[0]
        if len(first) < 24 and len(plain) > len(first):   # ("built; == oracle, and PINNED ...": one more clause)
            first = plain[:120]
        if len(first) > 120:
            first = first[:117].rsplit(" ", 1)[0] + " ..."
        name = re.split(r" \(`|: | `crates|; ", what)[0]
        if len(name) > 90:
            name = name[:87].rsplit(" ", 1)[0] + " ..."
        short.append("| %d | %s | %s | [details](docs/%s) |" % (k + 1, name, first, "SCOPEFILE#" + anchor))
    names = write_doc("scope", "Scope: rows (a) - (f) of SURVEY.md 8, as built", "The full cells of DESIGN.md section 0's table (round 5: DESIGN.md keeps one line per row).", scope_lines)
    # which file holds which row
    row_file = {}
    for nm in names:
        for m in re.finditer(r"^## Row (\d+)$", open(os.path.join(ROOT, "docs", nm + ".md")).read(), re.M):
            row_file[int(m.group(1))] = nm
    short = [s.replace("SCOPEFILE#row-%d" % (k + 1), "%s.md#row-%d" % (row_file[k + 1], k + 1)) for k, s in enumerate(short)]
    new0 = [h0, "", "One line per row; the full cells (what, where, state, tests) are in `docs/scope*.md`.", "", "| # | Row | State (first clause) | Full text |", "|---|---|---|---|"] + short + [""] + reflow([l for l in tail0 if l.strip()]) + [""]

    move(9, "aggregation", "Aggregation: the verifier circuit (f2, a5 / a6)")
    move(10, "segment_flow", "One statement per segment, one flow per task (f3 adapters, a2 / a3)")
    move(11, "one_key", "One aggregation key")
    move(12, "deferral", "Deferral (a6)")
    move(13, "shapes", "Per-proof chip presence: shapes")
    move(14, "configuration", "Configuration")
    move(15, "small_proofs_round4", "The small-proof path in round 4")
    move(8, "gaps", "Known gaps / next")
    # section 5: the kernel table and its notes move; DESIGN keeps a short table
    h5, body5 = sec[5]
    k_names = write_doc("kernels", "Kernels (gfx950), their bounds and algorithmic bytes", "Moved from DESIGN.md section 5 (round 5).", body5)
    new5 = [h5, "", "The prover is integer-VALU bound (the Poseidon2 row sponge), the transform passes are the one HBM-bound family.  Stage times of one 2^22 x 300 proof",
            "alone (88 ms; 79 ms per proof with three in flight); every figure, its source profile and the notes on each kernel: " + ", ".join("[docs/%s.md](docs/%s.md)" % (x, x) for x in k_names) + ".", "",
            "| Kernel(s) | ms | Bound | Algorithmic bytes |", "|---|---|---|---|",
            "| `k_hash_rows` (K2) | 44.6 | integer VALU (6.0 k instructions per permutation; 0.82 of its VALU floor) | 11.2 GB (traffic 11.4 GB) |",
            "| `k_ntt_pass4_ct` (K1) | 19.9 | HBM: two passes per transform, ~3 TB/s | 8 B per element per transform = 15.1 GB per LDE |",
            "| `quot_jit` (K5) | 6.3 | HBM / Infinity Cache (cells re-read per constraint class) | 11 GB (traffic 43 GB) |",
            "| `k_compress_layer(_coop)`, `k_compress_top` (K3) | 5.1 + 1.8 | permutation latency of the mid-size layers | 96 B per node |",
            "| `k_reduced_openings`, `k_col_reduce`, `k_open_finish` | 1.9 + 1.75 | VALU / HBM | 10.7 + 5.1 GB |",
            "| `k_hash_pairs`, `k_fri_fold` (K8) | 1.8 + 0.3 | VALU / HBM | 48 B per output |",
            "| transcript (T4), `k_grind` (K9) | 1.6 + 1.35 | latency (serial permutations) | - |", ""]
    index = ["## Further sections (moved to docs/ in round 5)", ""]
    for num, label in ((9, "Aggregation: the verifier circuit, leaf / internal nodes, TreeStream"), (10, "One statement per segment: the chips of a segment, executor, flow"),
                       (11, "One aggregation key whatever the depth"), (12, "Deferral: child proofs behind a parent guest's claims"), (13, "Shapes: per-proof chip presence"),
                       (14, "Configuration: zkhip_config, FlowOptions"), (15, "The small-proof path in round 4: what moved and what did not"), (8, "Known gaps / next")):
        index.append("* section %d -- %s: %s" % (num, label, ", ".join("[docs/%s.md](docs/%s.md)" % (x, x) for x in moved[num][0])))
    index += ["* the stale Merkle node of round 4 (cause, evidence, guards): [docs/stale_node.md](docs/stale_node.md)", ""]
    out = reflow(pre) + new0
    for num in (1, 2, 3, 4):
        out += [sec[num][0]] + reflow(sec[num][1])
    out += new5
    for num in (6, 7):
        out += [sec[num][0]] + reflow(sec[num][1])
    out += index
    open(path, "w").write("\n".join(out).rstrip("\n") + "\n")
    for f in sorted(os.listdir(os.path.join(ROOT, "docs"))):
        ls = open(os.path.join(ROOT, "docs", f)).read().split("\n")
        print("%-28s %4d lines, longest %d" % (f, len(ls), max(len(x) for x in ls)))
    ls = open(path).read().split("\n")
    print("DESIGN.md %d lines, longest %d" % (len(ls), max(len(x) for x in ls)))


if __name__ == "__main__":
    main(sys.argv[1])
