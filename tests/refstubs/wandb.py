"""Stand-in for wandb (absent from this image; the reference imports it at module level in smpl_sim/run.py and
agents/agent_humanoid.py and only calls it when cfg.no_log is false).  Test-side only: records what it is handed."""
logged = []
run = None


def init(**kw):
    global run
    import types
    run = types.SimpleNamespace(name=None, save=lambda: None, config=kw.get("config"))
    return run


def log(data=None, step=None, **kw):
    logged.append((step, data))
