"""``python hpmn.py <amazon|taobao|xlong>`` -- same CLI as the reference's code/hpmn.py
(/root/reference/code/hpmn.py:563-667), running on MI355X through hpmn_amd."""
import sys

from hpmn_amd.hpmn import Hpmn, Hpmn_Basic, Hpmn_Industry, main  # noqa: F401  (drop-in names)

if __name__ == "__main__":
    sys.exit(main(sys.argv))
