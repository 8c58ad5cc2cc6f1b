"""CPU-only: the C-ABI library loads and exports every symbol include/lrl_trace.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from conftest import REPO
from raytracing_amd import tracer


def declared_symbols():
    text = open(os.path.join(REPO, "include", "lrl_trace.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lrl_[a-z0-9_]+)\s*\(", text)))


def test_header_and_wrapper_agree():
    assert declared_symbols() == sorted(tracer.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(tracer.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), s
    assert L.lrl_version() == 2


def test_null_arguments_are_rejected_without_a_gpu():
    L = tracer.lib()
    assert L.lrl_create(0, None) == 1           # LRL_ERR_INVALID_ARG
    assert L.lrl_destroy(None) == 1
    assert L.lrl_last_error(None) == b"null context"
