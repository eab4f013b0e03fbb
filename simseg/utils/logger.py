"""Rank-aware stdout logger with the call surface the tools use: logger.info / emph / warning / error / debug
(mirror of simseg/utils/logger.py:55-139; root_only defaults to True)."""
import sys
import time

_COLORS = {"I": "", "E": "\033[0;32m", "W": "\033[0;33m", "X": "\033[0;31m", "D": "\033[0;36m"}
_RESET = "\033[0m"


STREAM = None      # None = stdout (the reference's logger prints); bench.py points it at stderr so that stdout carries its one JSON line only


def _emit(tag, args, root_only):
    from .context import ENV
    if root_only and ENV.rank != 0:
        return
    f = sys._getframe(2)
    where = f"{f.f_code.co_filename.rsplit('/', 1)[-1]}:{f.f_lineno}"
    stamp = time.strftime("%Y-%m-%d  %H:%M:%S")
    body = " ".join(str(a) for a in args)
    color = _COLORS[tag]
    print(f"{color}{'I' if tag == 'E' else tag if tag != 'X' else 'E'} {stamp} {where:<18} #{ENV.rank}] {body}{_RESET if color else ''}",
          flush=True, file=STREAM or sys.stdout)


def info(*args, root_only=True):
    _emit("I", args, root_only)


def emph(*args, root_only=True):
    _emit("E", args, root_only)


def warning(*args, root_only=True):
    _emit("W", args, root_only)


def error(*args, root_only=False):
    _emit("X", args, root_only)


def debug(*args, root_only=True):
    _emit("D", args, root_only)
