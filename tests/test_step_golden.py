"""NRLDPCDecoder.step() with state between calls -- incremental redundancy, sticky code-block CRC flags, CBGTI
(NRLDPCDecoder.m:236-239, 283-316, 336-339) -- against the committed fixture tests/golden/step_golden.npz.

CPU: the fixture is reproduced by the host mirror with the oracle as decoder core (pins fixture <-> oracle).
GPU: the host mirror (GPU decoder core, host-side state) and DeviceDecodeChain (every stage and all state on the
device, nrldpc_crc_check_harq_dev) both have to return the fixture's a_hat / ok / flags at every step."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
SCENARIOS = ("harq_jam", "harq_cbgti", "noharq")


def load(scenario):
    g = np.load(os.path.join(GOLD, "step_golden.npz"))
    plan = json.loads(bytes(g[scenario + "/plan"]).decode())
    steps = [(rv, cb, G, g["%s/g_tilde_%d" % (scenario, n)]) for n, (rv, cb, G) in enumerate(plan)]
    want = [(np.unpackbits(g["%s/a_hat_packed_%d" % (scenario, n)], axis=1)[:, :3842], g["%s/ok_%d" % (scenario, n)] != 0,
             g["%s/cb_pass_%d" % (scenario, n)]) for n in range(len(plan))]
    return g[scenario + "/a"], steps, want


def check(res, want, a, scenario):
    for n, ((a_hat, ok, passed), (wa, wok, wp)) in enumerate(zip(res, want)):
        assert (np.asarray(ok) == wok).all(), (scenario, n, ok, wok)
        assert (np.asarray(passed) == wp).all(), (scenario, n)
        assert (np.asarray(a_hat)[wok] == wa[wok]).all(), (scenario, n)   # where the reference returns [] a_hat is unspecified
    if scenario != "noharq":  # the scenario's point: success needs the state of earlier steps
        assert want[2][1][0] and (want[2][0][0] == a[0]).all() and not want[1][1][0]
        assert want[1][2][0, 0] == 1 and want[0][2][0].tolist() == [1, 0]


@pytest.mark.parametrize("scenario", SCENARIOS)
def test_fixture_is_what_the_oracle_backed_mirror_gives(pkg, orc, scenario):
    import make_step_golden as M
    a, steps, want = load(scenario)
    check(M.run(scenario, a, steps, M.oracle_decoder), want, a, scenario)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", SCENARIOS)
def test_host_mirror_step_reproduces_fixture(pkg, scenario):
    import make_step_golden as M
    a, steps, want = load(scenario)
    check(M.run(scenario, a, steps, pkg.NRLDPCDecoder), want, a, scenario)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", SCENARIOS)
def test_device_chain_state_reproduces_fixture(pkg, scenario):
    """b_hat_buffer, code_block_CRC_passed and CBGTI on the device (the device chain was stateless)."""
    import torch
    import make_step_golden as M
    DC = importlib.import_module("ldpc-3gpp-matlab_amd.device_chain")
    a, steps, want = load(scenario)
    p = pkg.NRLDPC(G=M.G_TX, **M.KW)
    chain = DC.DeviceDecodeChain(p, iterations=M.ITERS, I_HARQ=0 if scenario == "noharq" else 1, llr_dtype=np.float32)
    res = []
    for rv, cbgti, G, llr in steps:
        p.rv_id, p.CBGTI, p.G = rv, cbgti, G
        a_hat, ok, _ = chain.step(torch.from_numpy(llr.astype(np.float32)).cuda())
        res.append((a_hat.cpu().numpy(), ok.cpu().numpy(), chain.cb_pass.cpu().numpy()))
    check(res, want, a, scenario)
    if scenario == "noharq":                    # I_HARQ == 0: nothing is pending, a new batch size starts a new set
        chain.step(torch.zeros((2, p.G), dtype=torch.float32, device="cuda"))
    else:                                       # HARQ state pending: a different batch size needs an explicit reset()
        with pytest.raises(pkg.NRLDPCError):
            chain.step(torch.zeros((2, p.G), dtype=torch.float32, device="cuda"))
    chain.reset()
    a_hat, ok, _ = chain.step(torch.from_numpy(steps[0][3][:2].astype(np.float32)).cuda() if steps[0][2] == p.G
                              else torch.zeros((2, p.G), dtype=torch.float32, device="cuda"))
    assert ok.shape[0] == 2
    chain.close()
