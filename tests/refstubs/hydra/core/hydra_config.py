class HydraConfig:
    _cfg = None

    @classmethod
    def get(cls):
        return cls._cfg
