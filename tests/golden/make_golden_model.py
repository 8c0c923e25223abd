"""A checkpoint WRITTEN BY THE REFERENCE (Agent.save, agent.py:106-107) plus what the reference computes from it, for the
interchange test (agent.py:26-36): tests/golden/ref_model_dataeff.pth + ref_model_dataeff.npz.
Run in the build container (imports /root/reference):  python tests/golden/make_golden_model.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import agent as ref_agent  # noqa: E402  (the reference)

ARGS = dict(atoms=51, V_min=-10.0, V_max=10.0, batch_size=4, multi_step=3, discount=0.99, norm_clip=10.0, model=None,
            learning_rate=6.25e-5, adam_eps=1.5e-4, device=torch.device("cpu"), architecture="data-efficient",
            history_length=4, hidden_size=32, noisy_std=0.1)
ACTIONS = 4


def main():
    torch.manual_seed(20260924)
    np.random.seed(7)
    env = types.SimpleNamespace(action_space=lambda: ACTIONS)
    ag = ref_agent.Agent(types.SimpleNamespace(**ARGS), env)
    ag.reset_noise()
    ag.save(HERE, "ref_model_dataeff.pth")
    rs = np.random.RandomState(11)
    states = (rs.randint(0, 256, size=(6, 4, 84, 84)).astype(np.float32) / np.float32(255)).astype(np.float32)
    out = {"states_u8": np.rint(states * 255).astype(np.uint8)}
    ag.train()
    out["act_train"] = np.array([ag.act(torch.from_numpy(s)) for s in states], dtype=np.int64)
    out["q_train"] = np.array([ag.evaluate_q(torch.from_numpy(s)) for s in states], dtype=np.float32)
    ag.eval()
    out["act_eval"] = np.array([ag.act(torch.from_numpy(s)) for s in states], dtype=np.int64)
    out["q_eval"] = np.array([ag.evaluate_q(torch.from_numpy(s)) for s in states], dtype=np.float32)
    # a reloaded reference agent reproduces the same numbers (sanity of the fixture itself)
    ag2 = ref_agent.Agent(types.SimpleNamespace(**dict(ARGS, model=os.path.join(HERE, "ref_model_dataeff.pth"))), env)
    ag2.train()
    assert [ag2.act(torch.from_numpy(s)) for s in states] == list(out["act_train"])
    np.savez_compressed(os.path.join(HERE, "ref_model_dataeff.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "ref_model_dataeff.pth")), "bytes of checkpoint")


if __name__ == "__main__":
    main()
