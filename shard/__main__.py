from shardmerge_amd.__main__ import cli

if __name__ == "__main__":
    cli()
