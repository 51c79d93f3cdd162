"""Regenerate tools/README.md: one row per script (first paragraph of its docstring / leading comment) and the DESIGN.md / README sections that cite it.  CPU; run from the repository root."""
import ast, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
rows = []
for f in sorted(os.listdir("tools")):
    p = os.path.join("tools", f)
    if os.path.isdir(p):
        if f != "__pycache__":
            rows.append((f + "/", "probe kernels (standalone .hip + runner): " + ", ".join(sorted(x for x in os.listdir(p) if not x.startswith("__")))))
        continue
    if f == "README.md":
        continue
    txt, desc = open(p).read(), ""
    if f.endswith(".py"):
        try:
            desc = (ast.get_docstring(ast.parse(txt)) or "").strip().split("\n\n")[0].replace("\n", " ")
        except SyntaxError:
            pass
    else:
        desc = " ".join(l[1:].strip() for l in txt.split("\n")[1:6] if l.startswith("#"))
    rows.append((f, re.sub(r"\s+", " ", desc)[:260]))
design = open("DESIGN.md").read() + "\n## profiles/README\n" + open("profiles/README.md").read() + "\n## README\n" + open("README.md").read()
sec = re.compile(r"^## (\S+)", re.M)


def where(name):
    seen = []
    for m in re.finditer(re.escape(name), design):
        hs = list(sec.finditer(design[:m.start()]))
        if hs and hs[-1].group(1).rstrip(".") not in seen:
            seen.append(hs[-1].group(1).rstrip("."))
    return ", ".join(("section " + h) if h[0].isdigit() else h for h in seen[:4]) if seen else "-"


out = ["# tools/ -- measurement and A/B scripts (GPU box unless noted)\n",
       "Not product code: nothing under `naturaldiffusion_amd/` imports from here.  Every script names what it measures in its docstring; the DESIGN.md section (or README) that quotes its numbers is",
       "given where one does (`-` = used by another tool, the Makefile or a test only).  Development libraries (`-DNATINF_DEV`, tile-timeline stamps) are built as the csrc/Makefile header says",
       "and selected with `NATINF_LIB=<path to .so>`.  This file is generated: `python tools/index_tools.py`.\n", "| script | what it does | cited in |", "|---|---|---|"]
for f, d in rows:
    out.append("| `%s` | %s | %s |" % (f, d.replace("|", "/"), where(f.rstrip("/"))))
open("tools/README.md", "w").write("\n".join(out) + "\n")
print(len(rows), "entries")
