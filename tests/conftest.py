import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer oracle statistics run")


def pytest_collection_modifyitems(config, items):
    """device tests that are marked `slow` as well (the property-only sweep over all shipped parameter sets) only run when asked for:
    pytest -m "gpu and slow", or LART_RUN_SLOW=1 -- the default device suite has to stay within the driver's time limit"""
    if "slow" in (config.getoption("-m") or "") or os.environ.get("LART_RUN_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow device sweep: run with -m 'gpu and slow' or LART_RUN_SLOW=1")
    for it in items:
        if "slow" in it.keywords and "gpu" in it.keywords:
            it.add_marker(skip)


class Background:
    """a job started when the collection ends and joined by the fixture that needs it: the full-size BASELINE configurations of test_gpu_bigtau.py last as long as their
    longest photon history on a lone wave (~2 min of a mostly idle GPU), so they run on contexts and host threads of their own WHILE the other device tests
    run (the library keeps no state outside its contexts; its error strings are thread-local) and those tests come last in the order"""
    def __init__(self, fn):
        import threading
        self.result, self.error = None, None

        def work():
            try:
                self.result = fn()
            except BaseException as e:                       # noqa: BLE001 -- re-raised in join()
                self.error = e
        self.thread = threading.Thread(target=work, daemon=True); self.thread.start()

    def join(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.result


BACKGROUND = {}
import threading
STOP = threading.Event()                                     # set when the session ends: a background job still running (a test failed under -x) leaves its loop before the process exits


def pytest_sessionfinish(session, exitstatus):
    STOP.set()
    for job in BACKGROUND.values():
        job.thread.join(timeout=120)


def pytest_collection_finish(session):
    items = [it for it in session.items if "test_gpu_bigtau.py" in it.nodeid and not any(m.name == "skip" for m in it.iter_markers())]
    if not items or session.config.option.collectonly:
        return
    session.items[:] = [it for it in session.items if "test_gpu_bigtau.py" not in it.nodeid] + [it for it in session.items if "test_gpu_bigtau.py" in it.nodeid]
    import test_gpu_bigtau
    BACKGROUND["bigtau"] = Background(test_gpu_bigtau.run_all)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build(ref=False)
    return O


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def mueller_file(tmp_path_factory):
    """write the Mueller fixture back in the 3-header-line text layout that par%scatt_mat_file expects (setup.f90:597-627)"""
    z = np.load(os.path.join(GOLDEN, "mueller_Lyalpha.npz"))
    p = tmp_path_factory.mktemp("mueller") / "mueller_Lyalpha.dat"
    h, tab = z["header"], z["table"]
    with open(p, "w") as f:
        f.write("lambda(um), Cext(cm^2/H), albedo, <cos>, # of angles\n")
        f.write("  %.5f  %.5E  %.5f  %.5f  %d\n" % (h[0], h[1], h[2], h[3], len(tab)))
        f.write("cos(theta),     S11,               S12,             S33,             S34\n")
        for r in tab:
            f.write("%.4f  %.10E  %.10E  %.10E  %.10E\n" % tuple(r))
    return str(p)


def rms_diff(a, b):
    """spectrum rms-diff as BASELINE.json's metric defines it: sqrt(mean((a-b)^2)) / max(b)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.max(b))


def run_together(jobs, events=4096):
    """several simulations on one GPU at once: jobs = [(sim, nphotons, seed), ...]; every context has its own HIP stream and is driven by
    its own host thread (the C-ABI calls release the GIL; the library keeps no state outside its contexts), so the drain tails (the longest
    history of each job runs alone on a wave) overlap instead of adding up.  Same photon ids, same streams, same results as
    sim.run_simulation(n, seed=seed) one after the other."""
    import threading
    errors = []

    def drive(sim, n, seed):
        try:
            sim.ctx.begin(int(n), first_id=1, stride=1, seed=int(seed)); sim.nphotons_run = int(n)
            while True:
                sim.ctx.step(1, events)
                if sim.ctx.sync()[1] == 0 or STOP.is_set():
                    break
        except Exception as e:                               # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=drive, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


@pytest.fixture(scope="session")
def h2_data_dir(tmp_path_factory):
    """par%h2_data_dir: the X-state level table h2_init reads (h2_mod.f90:250-296), written back from tests/golden/h2_energy_X.npz in the layout of
    the reference's data/h2/energy_X.dat (a magic-number first line, '#' comments, rows 'V J E[cm^-1]')"""
    z = np.load(os.path.join(GOLDEN, "h2_energy_X.npz"))
    d = tmp_path_factory.mktemp("h2")
    with open(d / "energy_X.dat", "w") as f:
        f.write("2 4 29 // magic number\n#H2 electronic ground state rovibrational level energies [cm^-1]\n")
        for v, j, e in zip(z["V"], z["J"], z["E_cm"]):
            f.write("%d\t%d\t%.17g\n" % (int(v), int(j), float(e)))
    return str(d)
