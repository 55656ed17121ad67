"""A reader for the re-authored rule files (rules/*.smk, eval_variant_custom.smk) -- no Snakemake exists in the build image.

It does what Snakemake does with the parts of the language those files use, and nothing more: the top-level Python of a file
runs in a namespace the test provides (`config`, the names the including workflow defines), `configfile:` is a no-op (the
test passes `config`), every `rule NAME:` becomes an entry with its `input` / `output` / `params` sections EVALUATED (keyword
and positional items, `expand`, `rules.<name>.output.<item>`), its `threads`, and its `run:` body COMPILED; `run_rule` executes
that body with `input` / `output` / `params` / `wildcards` / `threads` bound, the way Snakemake does.  `shell:` rules are kept
as text."""
import itertools
import re
import textwrap
import types


class Items(list):
    """Snakemake's Namedlist in miniature: positional items in order, named ones also as attributes."""

    def __init__(self, args=(), kwargs=None):
        super().__init__()
        self._names = {}
        for a in args:
            self._add(a)
        for k, v in (kwargs or {}).items():
            self._names[k] = v
            self._add(v)

    def _add(self, v):
        if isinstance(v, (list, tuple)):
            self.extend(v)
        else:
            self.append(v)

    def __getattr__(self, k):
        try:
            return self.__dict__["_names"][k]
        except KeyError:
            raise AttributeError(k)

    def get(self, k, default=None):
        return self._names.get(k, default)

    def keys(self):
        return self._names.keys()


def expand(pattern, **lists):
    """every combination of the keyword lists, first keyword slowest (Snakemake's order)"""
    keys = list(lists)
    vals = [v if isinstance(v, (list, tuple)) else [v] for v in lists.values()]
    return [pattern.format(**dict(zip(keys, combo))) for combo in itertools.product(*vals)]


_RULE = re.compile(r"^rule\s+(\w+)\s*:\s*$")
_SECTION = re.compile(r"^    (\w+)\s*:\s*(.*)$")


def load(path, namespace):
    """-> (namespace after the file's top-level code, {rule name: {"input", "output", "params": Items, "threads", "run": code, "shell": str}})"""
    lines = open(path).read().split("\n")
    top, rules_src, i = [], [], 0
    while i < len(lines):
        m = _RULE.match(lines[i])
        if m:
            j = i + 1
            while j < len(lines) and (lines[j].startswith("    ") or not lines[j].strip()):
                j += 1
            rules_src.append((m.group(1), len(top), lines[i + 1:j]))
            i = j
            continue
        if re.match(r"^(configfile|include|ruleorder)\s*:", lines[i]):
            i += 1
            continue
        top.append(lines[i])
        i += 1
    ns = dict(namespace)
    ns.setdefault("expand", expand)
    rules_obj = types.SimpleNamespace()
    ns["rules"] = rules_obj
    out = {}
    done = 0
    for name, upto, body in rules_src:       # the top-level code in front of a rule runs before the rule is read
        exec(compile("\n".join(top[done:upto]), path, "exec"), ns)
        done = upto
        sections, cur = {}, None
        for ln in body:
            m = _SECTION.match(ln)
            if m and not ln.startswith("        "):
                cur = m.group(1)
                sections[cur] = [m.group(2)] if m.group(2) else []
            elif cur is not None:
                sections[cur].append(ln)
        rule = {"name": name}
        for sec in ("input", "output", "params"):
            text = textwrap.dedent("\n".join(sections.get(sec, []))).strip()
            text = "\n".join(l for l in text.split("\n") if not l.strip().startswith("#"))
            args, kwargs = eval("(lambda *a, **k: (a, k))(%s)" % text, ns) if text else ((), {})
            rule[sec] = Items(args, kwargs)
        rule["threads"] = eval("\n".join(sections["threads"]).strip(), ns) if "threads" in sections else 1
        if "run" in sections:
            rule["run"] = compile(textwrap.dedent("\n".join(sections["run"])), "%s:%s" % (path, name), "exec")
        if "shell" in sections:
            rule["shell"] = textwrap.dedent("\n".join(sections["shell"]))
        setattr(rules_obj, name, types.SimpleNamespace(input=rule["input"], output=rule["output"], params=rule["params"]))
        out[name] = rule
    exec(compile("\n".join(top[done:]), path, "exec"), ns)
    return ns, out


def run_rule(ns, rule, wildcards=None):
    """execute the rule's `run:` body as Snakemake would"""
    env = dict(ns)
    env.update(input=rule["input"], output=rule["output"], params=rule["params"], threads=rule["threads"],
               wildcards=types.SimpleNamespace(**(wildcards or {})))
    exec(rule["run"], env)
    return env
