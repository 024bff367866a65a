"""Test double for the scheduler's event backend, written against the PRODUCT's extension point (bfh_obs_create_custom /
bfh_event_ops, include/dsabf_host.h): events complete -- or fail -- only when the test says so.  Lives in tests/ only;
libdsabf.so contains no test scaffolding."""
import ctypes as C

from dsabeamformer_amd._lib import BF_ERR_DEVICE, BF_NOT_READY, BF_OK, BfhEventOps


class FakeEvents:
    def __init__(self):
        self.state = {}            # id -> 0 never recorded (queries "done", like a fresh CUDA event), 1 pending, 2 done
        self.next_id = 1
        self.pending_transfers, self.pending_analyses = [], []
        self.fail_query = self.fail_record = self.fail_create = False
        self.ops = BfhEventOps(None, BfhEventOps.CREATE(self._create), BfhEventOps.DESTROY(self._destroy),
                               BfhEventOps.RECORD(self._record_transfer), BfhEventOps.RECORD(self._record_analysis),
                               BfhEventOps.QUERY(self._query))

    def _create(self, _user):
        if self.fail_create:
            return None
        i = self.next_id
        self.next_id += 1
        self.state[i] = 0
        return i

    def _destroy(self, _user, ev):
        self.state.pop(ev, None)
        self.pending_transfers = [e for e in self.pending_transfers if e != ev]
        self.pending_analyses = [e for e in self.pending_analyses if e != ev]

    def _record(self, ev, queue):
        if self.fail_record:
            return BF_ERR_DEVICE
        self.state[ev] = 1
        queue.append(ev)
        return BF_OK

    def _record_transfer(self, _user, ev):
        return self._record(ev, self.pending_transfers)

    def _record_analysis(self, _user, ev):
        return self._record(ev, self.pending_analyses)

    def _query(self, _user, ev):
        if self.fail_query:
            return BF_ERR_DEVICE
        return BF_NOT_READY if self.state.get(ev) == 1 else BF_OK

    def complete(self, n_transfers=0, n_analyses=0):
        for _ in range(min(n_transfers, len(self.pending_transfers))):
            self.state[self.pending_transfers.pop(0)] = 2
        for _ in range(min(n_analyses, len(self.pending_analyses))):
            self.state[self.pending_analyses.pop(0)] = 2


def make_obs(host, cfg, debug=True, **kw):
    """ObservationLoopState on fake events; obs.fake_complete(nt, na) completes the oldest pending events."""
    fake = FakeEvents()
    obs = host.ObservationLoopState(cfg, debug=debug, event_ops=fake.ops, **kw)
    obs.fake = fake
    obs.fake_complete = fake.complete
    return obs
