"""PAO virtual localisation: the reference's driver refuses it
("PAO not yet fully implemented.", nbed/driver.py:819-820); the name stays importable."""


class PAOLocalizer:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("PAO not yet fully implemented.")
