"""The two host mailboxes of the polyhedron engine (Mail: one-cut pipeline, RState: rounds of independent cuts) are read
through checksums keyed by the sequence number, because a new sequence number was once seen on the host before its content
("A torn mailbox read").  Host-only: a writer thread publishes the number FIRST and the content afterwards, word
by word; the readers (the very functions wait_mail / r2_wait spin on) must never hand out foreign content."""
import ctypes

import pytest

from bensolve_amd._lib import load_library


@pytest.mark.parametrize("which,name", [(0, "Mail"), (1, "RState")])
def test_mailbox_readers_reject_a_sequence_number_that_arrives_before_its_content(which, name):
    lib = load_library()
    lib.bslv_selftest_mailbox.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    torn = ctypes.c_long()
    rc = lib.bslv_selftest_mailbox(which, 5000, ctypes.byref(torn))
    assert rc == 0, "%s: %d messages were accepted with the content of another sequence number" % (name, -1 - rc)
    # the writer leaves a window after every sequence number: the reader must have had to look again at least sometimes
    assert torn.value > 0, "%s: the adversarial writer never produced a torn read -- the test did not test anything" % name
