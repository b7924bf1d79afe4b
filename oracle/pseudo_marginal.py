"""
TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain restatement of one lock-step random-walk
Rosenbluth-Metropolis-Hastings step with auxiliary variables, written per chain with Python loops and INJECTED draws
(proposal increments and acceptance uniforms are arguments), after

    src/rodeo/inference/pseudo_marginal.py:160-172 (additive step), :342-379 (kernel), :438-449 (energies),
    :469-483 (generate: propose -> log-density at the proposal -> acceptance ratio -> binomial sampling)

and blackjax's ``proposal.compute_asymmetric_acceptance_ratio`` / ``static_binomial_sampling`` (third-party, not under
/root/reference; blackjax <= 1.2.3 on py < 3.10, else unpinned -- pyproject.toml:35-36), whose published rule is
    log_p = (-logdensity_old [- q(old | new)]) - (-logdensity_new [- q(new | old)]),  NaN -> -inf,
    p = min(1, exp(log_p)),  accept iff u < p.
**Parity with blackjax's bit-stream is unpinned** (SURVEY.md section 8c): there is no reference fixture for it.
"""
import math
import numpy as np


def rmh_step(position, logdensity, auxdata, increments, uniforms, logdensity_at, proposal_logdensity=None):
    """
    position (C, dim), logdensity (C,), auxdata list of length C (or None), increments (C, dim), uniforms (C,);
    ``logdensity_at(c, x) -> (value, aux)`` evaluates chain c's (possibly noisy) log-density at x;
    ``proposal_logdensity(x_a, x_b)`` optional, the reference's ``proposal_logdensity_fn(new_state, prev_state)`` with
    positions for states (pseudo_marginal.py:446-447): the energy of a move prev -> new is
    ``-logdensity(new) - proposal_logdensity(new, prev)``, so for a correct Metropolis-Hastings ratio it must return
    ``log q(prev | new)`` (for independent proposals: ``log q(prev)``).
    Returns (position, logdensity, auxdata, accepted, p_accept).
    """
    C = len(position)
    pos_out, ld_out, aux_out = [], [], []
    acc, pacc = np.zeros(C, dtype=bool), np.zeros(C)
    for c in range(C):
        x_old = np.asarray(position[c], dtype=float)
        x_new = x_old + np.asarray(increments[c], dtype=float)
        ld_new, aux_new = logdensity_at(c, x_new)
        e_back = -float(logdensity[c])            # energy of the move new -> old
        e_fwd = -float(ld_new)                    # energy of the move old -> new
        if proposal_logdensity is not None:
            e_back -= proposal_logdensity(x_old, x_new)     # transition_energy(prev = new, new = old)
            e_fwd -= proposal_logdensity(x_new, x_old)      # transition_energy(prev = old, new = new)
        log_p = e_back - e_fwd
        if math.isnan(log_p):
            log_p = -math.inf
        p = 1.0 if log_p >= 0 else math.exp(log_p)
        a = bool(uniforms[c] < p)
        acc[c], pacc[c] = a, p
        pos_out.append(x_new if a else x_old)
        ld_out.append(float(ld_new) if a else float(logdensity[c]))
        aux_out.append(aux_new if a else (None if auxdata is None else auxdata[c]))
    return np.array(pos_out), np.array(ld_out), aux_out, acc, pacc
