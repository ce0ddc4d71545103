"""host/lib.py lockstep: the pairing of two segments' groupable calls, on ONE thread (two greenlets; round 6) and on two threads -- the launch log must be the
same function of the two call sequences either way.  No GPU: the C entry points, the group recorder and the stream handle are replaced by recorders."""
import pytest

import magic_amd  # noqa: F401
from magic_amd.host import lib as L


class _FakeLib:
    def __init__(self, log):
        self.log = log

    def magic_group_begin(self):
        self.log.append("[")
        return 0

    def magic_group_end(self, stream):
        self.log.append("]")
        return 0


@pytest.fixture
def recorded(monkeypatch):
    log = []
    monkeypatch.setattr(L, "load", lambda: _FakeLib(log))
    monkeypatch.setattr(L, "_fn", lambda name: (lambda *a: (log.append((name,) + a), 0)[1]))
    monkeypatch.setattr(L, "_raw_call", lambda name, args: log.append(("solo", name) + tuple(args)))
    monkeypatch.setattr(L, "stream", lambda: 0)
    # the threaded form binds its helper thread to the caller's device and stream: stand-ins, there is no GPU here
    import contextlib

    import torch
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    return log


def _seg(tag, calls):
    def run():
        for c in calls:
            if c == "boom":
                raise ValueError(tag)
            L.call(c, tag)
        return tag
    return run


A = ["magic_gemm", "magic_ln_fwd", "magic_cast", "magic_gemm", "magic_attn_fwd"]          # magic_cast is not groupable: launched where it stands
B = ["magic_gemm", "magic_ln_fwd", "magic_gemm", "magic_attn_fwd", "magic_gemm", "magic_gemm"]


def _run(form, monkeypatch, log, a=A, b=B):
    monkeypatch.setattr(L, "LOCKSTEP_FORM", form)
    del log[:]
    out = L.lockstep(_seg("a", a), _seg("b", b))
    return out, list(log)


def test_one_thread_pairs_twin_calls_and_finishes_the_longer_segment_alone(recorded, monkeypatch):
    if L._greenlet is None:
        pytest.skip("no greenlet module")
    out, log = _run("greenlets", monkeypatch, recorded)
    assert out == ("a", "b")
    groups = "".join(x if isinstance(x, str) else "." for x in log)
    assert groups.count("[") == 4                       # gemm+gemm, ln+ln, gemm+gemm, attn+attn
    assert ("solo", "magic_cast", "a") in log
    assert log[-2:] == [("solo", "magic_gemm", "b"), ("solo", "magic_gemm", "b")]      # b's tail after a ended
    inside = [log[i + 1:i + 3] for i, x in enumerate(log) if x == "["]
    assert all(p[0][0] == p[1][0] and {p[0][1], p[1][1]} == {"a", "b"} for p in inside), inside
    assert getattr(L._tls, "lockstep", None) is None


def test_one_thread_and_two_threads_issue_the_same_launches(recorded, monkeypatch):
    if L._greenlet is None:
        pytest.skip("no greenlet module")
    _, one = _run("greenlets", monkeypatch, recorded)
    _, two = _run("threads", monkeypatch, recorded)
    assert one == two
    _, one = _run("greenlets", monkeypatch, recorded, a=B, b=A[:2])
    _, two = _run("threads", monkeypatch, recorded, a=B, b=A[:2])
    assert one == two


@pytest.mark.parametrize("form", ["greenlets", "threads"])
def test_an_error_in_one_segment_reaches_the_caller_and_leaves_no_lockstep_behind(recorded, monkeypatch, form):
    if form == "greenlets" and L._greenlet is None:
        pytest.skip("no greenlet module")
    monkeypatch.setattr(L, "LOCKSTEP_FORM", form)
    with pytest.raises(ValueError):
        L.lockstep(_seg("a", ["magic_gemm", "magic_gemm", "boom"]), _seg("b", B))
    with pytest.raises(ValueError):
        L.lockstep(_seg("a", B), _seg("b", ["magic_gemm", "boom"]))
    assert getattr(L._tls, "lockstep", None) is None
    out, _ = _run(form, monkeypatch, recorded)           # and the next one works
    assert out == ("a", "b")
