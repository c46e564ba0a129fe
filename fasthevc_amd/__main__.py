from .build import build_hip

if __name__ == "__main__":
    print(build_hip(verbose=True))
