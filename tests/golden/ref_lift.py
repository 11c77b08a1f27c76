"""Runs definitions of the reference IN PLACE, for fixture generation in the build container only.

Most reference modules cannot be imported here: their module level pulls in faiss / torchvision / cv2 /
librosa / ``models.lstm``.  The class and function bodies on the hot path, however,
need nothing but torch / numpy / scipy.  ``lift`` parses a reference source file with ``ast``, picks the
named top-level definitions (or ``Class.method`` members) out of the tree, compiles exactly those nodes --
from the file where it lies under /root/reference, nothing is copied or stubbed -- and executes them in a
namespace that provides the third-party names the file itself imports (torch, nn, F, np, signal, Variable,
dist, math).  The objects that come back ARE the reference's code.

Nothing here is imported by tests, the product or the GPU box: only ``make_ref_goldens.py`` uses it, and
only arrays leave it.
"""
import ast
import math

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F
from scipy import signal
from torch.autograd import Variable

REF = "/root/reference"


def base_namespace(**extra):
    ns = dict(torch=torch, nn=nn, F=F, np=np, numpy=np, signal=signal, Variable=Variable, dist=dist, math=math,
              device=torch.device("cpu"))
    ns.update(extra)
    return ns


def lift(relpath, names, ns=None):
    """Executes the definitions ``names`` of /root/reference/<relpath>; returns the namespace.

    A name is a top-level ``def``/``class`` (first occurrence in the file), or ``Class.method`` for one method
    of a class that cannot be instantiated here (returned as a plain function under the key ``Class.method``).
    """
    path = f"{REF}/{relpath}"
    tree = ast.parse(open(path).read(), filename=path)
    ns = base_namespace() if ns is None else ns
    ns.setdefault("__builtins__", __builtins__)
    top = {}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name not in top:
            top[node.name] = node
    for name in names:
        if "." in name:
            cls, meth = name.split(".")
            node = next(n for n in top[cls].body if isinstance(n, ast.FunctionDef) and n.name == meth)
            node.decorator_list = []
            scratch = dict(ns)
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scratch)
            fn = scratch[meth]
            # the function's globals must be the shared namespace (it may call other lifted names)
            import types
            ns[name] = types.FunctionType(fn.__code__, ns, fn.__name__, fn.__defaults__, fn.__closure__)
        else:
            exec(compile(ast.Module(body=[top[name]], type_ignores=[]), path, "exec"), ns)
    return ns
