import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import expect  # noqa: E402

# Budget of the GPU suite on the driver's box: the step is killed at 900 s; the suite
# prints its own wall time against this figure at the end of the run.
GPU_SUITE_BUDGET_S = 450


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._sgfhe_t0 = time.time()


def pytest_collection_modifyitems(config, items):
    # Recording the expectations (tests/expect.py) runs the oracle halves of the GPU tests on the CPU:
    # only tests written for it (they take the `exp` fixture) can run without an engine.
    if expect.RECORD:
        keep = [it for it in items if "exp" in getattr(it, "fixturenames", ())]
        drop = [it for it in items if it not in keep]
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep


_EXP = None


@pytest.fixture(scope="session")
def exp():
    """Recorded oracle expectations of the fixed-seed GPU comparisons (tests/expect.py)."""
    global _EXP
    if _EXP is None:
        _EXP = expect.Expect()
    return _EXP


def pytest_sessionfinish(session, exitstatus):
    if _EXP is not None:
        _EXP.save()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    tr = terminalreporter
    if _EXP is not None:
        if _EXP.record:
            tr.write_line("expectations recorded: %d digests -> %s" % (len(_EXP.new), _EXP.path))
        else:
            tr.write_line("oracle expectations: %d served from tests/golden/gpu_expect.json, %d computed live%s"
                          % (_EXP.served, len(_EXP.computed),
                             (" (" + ", ".join(_EXP.computed[:8]) + (" ..." if len(_EXP.computed) > 8 else "") + ")")
                             if _EXP.computed else ""))
    mark = config.getoption("-m") or ""
    if "gpu" in mark and "not gpu" not in mark and not expect.RECORD:
        dt = time.time() - config._sgfhe_t0
        tr.write_line("GPU suite wall time %.0f s (budget %d s, driver limit 900 s)%s"
                      % (dt, GPU_SUITE_BUDGET_S, "" if dt <= GPU_SUITE_BUDGET_S else "  ** OVER BUDGET **"))
        # the slowest items, whatever flags the runner was started with (the driver's command has no --durations):
        # the next test that eats the budget shows up in the round's log
        reps = [r for rs in tr.stats.values() for r in rs
                if hasattr(r, "duration") and getattr(r, "when", None) in ("setup", "call")]
        for r in sorted(reps, key=lambda r: -r.duration)[:15]:
            tr.write_line("  %7.2f s %-5s %s" % (r.duration, r.when, r.nodeid))


@pytest.fixture(scope="session")
def S():
    import sgfhe_jl_amd
    return sgfhe_jl_amd


@pytest.fixture(scope="session")
def oc():
    import oracle_c
    oracle_c.build()
    return oracle_c


def oracle_threads():
    """Threads for the oracle's OpenMP loops: the cgroup CPU quota where there is one (the GPU boxes show
    256 cores and a quota of 16), else the cores."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(q) // int(per))
    except OSError:
        pass
    return min(32, os.cpu_count() or 1)


class GpuKeys:
    """One engine per Params(n) for the whole session, keyed on the device from fixed seeds (secret key
    seed 21, key seed 22: the key of tests/golden/p1024*.json at n = 1024), with the oracle's objects for
    the same seeds made on demand.  Tests that only need "an engine with a valid key" share these instead
    of generating the key again (Params(1024): 1.3 s on the device, 10 s in the oracle)."""

    SK_SEED, KEY_SEED = 21, 22

    def __init__(self, S, oc):
        self.S, self.oc = S, oc
        self._eng, self._o, self._khat = {}, {}, {}

    def oracle(self, n):
        if n not in self._o:
            o = self.oc.Oracle.from_params(self.S.Params(n))
            self._o[n] = (o, o.private_key(self.SK_SEED))
        return self._o[n]

    def engine(self, n):
        """(params, oracle, secret key, engine); the engine in its default state is the caller's to leave so."""
        params = self.S.Params(n)
        o, sk = self.oracle(n)
        if n not in self._eng:
            eng = self.S.Engine(params)
            eng.generate_key(sk, self.KEY_SEED)
            self._eng[n] = eng
        return params, o, sk, self._eng[n]

    def khat(self, n):
        """The oracle's key for the same seeds in the NTT domain (bootstrap_batch(..., opt=True))."""
        if n not in self._khat:
            o, sk = self.oracle(n)
            T = oracle_threads()
            self._khat[n] = o.key_transform(o.bootstrap_key(sk, self.KEY_SEED, threads=T), threads=T)
        return self._khat[n]

    def close(self):
        for e in self._eng.values():
            e.close()
        self._eng.clear()
        self._khat.clear()


@pytest.fixture(scope="session")
def gpu_keys(S, oc):
    k = GpuKeys(S, oc)
    yield k
    k.close()
