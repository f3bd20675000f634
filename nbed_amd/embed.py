"""Import interface (mirror of nbed/embed.py:39-86): ``nbed(config | path | None, **kwargs)``."""

from __future__ import annotations

import argparse
import json
from pathlib import Path

from .config import NbedConfig, parse_config
from .driver import NbedDriver


def nbed(config: NbedConfig | str | Path | None = None, provider=None, backend=None,
         hamiltonian_format: str = "dense", **config_kwargs) -> NbedDriver:
    """Validate the configuration, run ``NbedDriver.embed()`` and return the driver."""
    driver = NbedDriver(parse_config(config, **config_kwargs), provider=provider, backend=backend,
                        hamiltonian_format=hamiltonian_format)
    driver.embed()
    return driver


def cli() -> None:
    """``nbed --config file.json`` (nbed/utils.py:52-77)."""
    parser = argparse.ArgumentParser(description="Projection-based embedding on MI355X.")
    parser.add_argument("--config", "-c", type=str, required=True, help="path to a .json config file")
    args = parser.parse_args()
    with open(args.config) as fh:
        nbed(NbedConfig(**json.load(fh)))


if __name__ == "__main__":
    cli()
