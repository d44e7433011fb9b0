"""(thread, device) lookup of the stateless matchers' workspaces: compiled with plain g++ and run on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workspace_lookup_is_per_thread_and_device(tmp_path):
    exe = str(tmp_path / "workspace_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-fsanitize=address,undefined",
                    "-I" + os.path.join(ROOT, "orb_slam2_map_amd", "csrc"), os.path.join(ROOT, "tests", "workspace_test.cpp"),
                    "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "workspace_test ok" in r.stdout, r.stdout


def test_stateless_entry_points_use_the_lookup():
    """No entry point keeps a bare thread_local workspace object any more."""
    for name in ("matcher_bf.hip", "matcher_proj.hip"):
        src = open(os.path.join(ROOT, "orb_slam2_map_amd", "csrc", name)).read()
        assert "per_device_workspace<" in src
        assert "static thread_local Ws ws" not in src and "static thread_local ProjWorkspace ws" not in src
        assert "static thread_local bool attr_set" not in src  # function attributes are per device as well


def test_id_hash_and_pointer_index(tmp_path):
    """The MapPoint table's host-side id -> row hash (insert, lookup, growth, the rollback of a refused call) and the shim's
    pointer index against std::map / a linear search, under AddressSanitizer + UBSan."""
    exe = str(tmp_path / "id_hash_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                    "-I" + os.path.join(ROOT, "orb_slam2_map_amd", "csrc"), "-I" + os.path.join(ROOT, "orb_slam2_map_amd", "shim"),
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "id_hash_test.cpp"), "-o", exe, "-pthread"],
                   check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "id_hash ok" in r.stdout and "ptr_index ok" in r.stdout, r.stdout
