"""CLI contract of the reference (main.rs:55-61): `prog <scene.json> <out.png>`."""
import sys

from .api import deploy_render


def main(argv):
    if len(argv) != 3:
        print("usage: python -m rs_ray_toy_amd <scene.json> <out.png>", file=sys.stderr)
        return 2
    deploy_render(argv[1], argv[2])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
